#!/usr/bin/env python3
"""ORACLE — TEST INFRASTRUCTURE.  Golden vectors for the lz77 flavour on blocks above 64 KiB (WINDOW_BITS 16 with a window
that really slides): runs the REAL reference (oracle/_ref/liblz77_w16.so, compiled from /root/reference by oracle/Makefile)
and the restatement on seeded synthetic inputs, refuses to write on any mismatch, and records bit counts + SHA-256 of
the streams in tests/golden/lz77_wide.json.  Inputs are regenerated from their seeds by the tests (only hashes are kept).
Build container only (needs oracle/_ref)."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from compression_algorithms_amd import synth  # noqa: E402
from oracle import orc, ref  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def case(data, wbits=16):
    if orc.past_table_end(data, wbits, wbits + 6, False):
        return None                                   # the reference would write past its bucket array: not pinned
    r, nb = ref.lz77_compress(data, wbits)
    o, nb2 = orc.lz77_encode(data, wbits, 4)
    if nb != nb2 or not np.array_equal(r, o):
        raise SystemExit("RESTATEMENT != REFERENCE")
    return {"bits": int(nb), "sha256": sha(r)}


def main():
    assert ref.available(), "run `make -C oracle ref` first"
    out = {}
    for name, seed, n, block in (("enwik_1MiB", 41, 1 << 20, 1 << 20), ("enwik_600k_b256k", 42, 600_000, 262144),
                                 ("enwik_300k_b128k", 43, 300_000, 131072)):
        data = synth.enwik_like(n, seed=seed).numpy()
        e = {"seed": seed, "n": n, "block": block, "input_sha256": sha(data), "blocks": []}
        for at in range(0, n, block):
            e["blocks"].append(case(data[at:at + block].tobytes()))
        out[name] = e
        print(name, [b and b["bits"] for b in e["blocks"]])
    with open(os.path.join(ROOT, "tests", "golden", "lz77_wide.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
