#!/usr/bin/env python3
"""ORACLE — TEST INFRASTRUCTURE.  Golden vectors for the reference's brute-force parser (lz77_compress_old,
algorithms/lz77/lz77.c:185-262): runs the REAL reference (oracle/_ref/liblz77_w{14,16}.so, compiled from /root/reference by
oracle/Makefile) and the restatement (orc_lz77_old_encode) on seeded inputs, refuses to write on any mismatch, and records
bit counts + SHA-256 of the streams in tests/golden/lz77_old.json.  The tests regenerate the inputs from `make_input`
below (only hashes are kept).  Build container only (needs oracle/_ref)."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from compression_algorithms_amd import synth  # noqa: E402
from oracle import orc  # noqa: E402

CASES = [                        # name, kind, n, seed, wbits
    ("enwik_50k_w14", "enwik", 50_000, 61, 14),
    ("enwik_below_window", "enwik", 16_000, 62, 14),       # shorter than the window: every token a literal (the unsigned wrap)
    ("enwik_16383", "enwik", 16_383, 63, 14),
    ("enwik_16384", "enwik", 16_384, 64, 14),
    ("enwik_16390", "enwik", 16_390, 65, 14),              # the last tokens' matches are cut by the end of the buffer
    ("zeros_40k", "zeros", 40_000, 0, 14),
    ("period3_40k", "period3", 40_000, 0, 14),
    ("period16383_50k", "period16383", 50_000, 0, 14),     # the only copy sits at the far end of the window
    ("period16384_50k", "period16384", 50_000, 0, 14),     # ... and just outside it
    ("random_40k", "random", 40_000, 1, 14),
    ("lowent_40k", "lowent", 40_000, 2, 14),
    ("enwik_120k_w16", "enwik", 120_000, 66, 16),
    ("empty", "zeros", 0, 0, 14),
    ("one_byte", "random", 1, 3, 14),
]


def make_input(kind, n, seed):
    if kind == "enwik":
        return synth.enwik_like(n, seed=seed).numpy()
    if kind == "zeros":
        return np.zeros(n, np.uint8)
    if kind == "random":
        return np.random.default_rng(seed).integers(0, 256, n, dtype=np.uint8)
    if kind == "lowent":
        return np.random.default_rng(seed).integers(97, 101, n, dtype=np.uint8)
    if kind.startswith("period"):
        k = int(kind[6:])
        base = np.random.default_rng(k).integers(0, 256, k, dtype=np.uint8)
        return np.resize(base, n)
    raise ValueError(kind)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    from oracle import ref
    assert ref.available(), "run `make -C oracle ref` first"
    out = {}
    for name, kind, n, seed, wbits in CASES:
        data = make_input(kind, n, seed)
        o, nb = orc.lz77_old_encode(data, wbits, 4)
        r, nb2 = ref.lz77_compress_old(data, wbits)
        if nb != nb2 or not np.array_equal(r, o):
            raise SystemExit(f"RESTATEMENT != REFERENCE on {name}")
        if n and not np.array_equal(orc.lz77_decode(o, nb, n, wbits, 4), data):
            raise SystemExit(f"round trip failed on {name}")
        out[name] = {"kind": kind, "n": n, "seed": seed, "wbits": wbits, "input_sha256": sha(data), "bits": int(nb), "sha256": sha(r)}
        print(name, nb)
    with open(os.path.join(ROOT, "tests", "golden", "lz77_old.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
