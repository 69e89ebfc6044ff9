#!/usr/bin/env python3
"""ORACLE — TEST INFRASTRUCTURE ONLY.

tests/golden/defh.json: fixtures for deflate "mode H" (oracle/orc_defh.c).

PARITY UNPINNED for the bit stream itself — the reference stops at a TODO where this stage would be
(algorithms/deflate/lz77.c:279), so these vectors are the ORACLE's output, committed so that a change of the
format shows up as a diff.  What ties them to the reference: the token streams that are coded here are
the ones tests/golden/kat_small.json and enwik_like_300k.json pin to the real reference (checked below before
anything is written), and the length procedure is cross-checked in tests/test_oracle_defh.py against the
reference-pinned byte Huffman coder.

    python oracle/gen_golden_defh.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def records(data, block=65536):
    d = orc.Deflate(block)
    toks, recs = [], []
    for at in range(0, len(data), block):
        d.fresh()
        t = d.block_encode(data[at:at + block])
        toks.append(t)
        recs.append(orc.defh_encode_block(t))
    return toks, recs


def main():
    out = {}
    kat = json.load(open(os.path.join(GOLD, "kat_small.json")))
    small = {}
    for name, e in kat.items():
        data = np.frombuffer(bytes.fromhex(e["input_hex"]), dtype=np.uint8)
        toks, recs = records(data)
        tok_hex = b"".join(t.tobytes() for t in toks).hex()
        if tok_hex != e["deflate_fresh_hex"]:
            sys.exit(f"{name}: oracle tokens differ from the reference golden vector; not writing fixtures")
        small[name] = b"".join(r.tobytes() for r in recs).hex()
    out["kat_small"] = small
    sample = np.fromfile(os.path.join(GOLD, "enwik_like_300k.bin"), dtype=np.uint8)
    g = json.load(open(os.path.join(GOLD, "enwik_like_300k.json")))["deflate_independent"]
    toks, recs = records(sample)
    if hashlib.sha256(b"".join(t.tobytes() for t in toks)).hexdigest() != g["sha256"]:
        sys.exit("300k sample: oracle tokens differ from the reference golden vector; not writing fixtures")
    stream = b"".join(r.tobytes() for r in recs)
    out["enwik_like_300k"] = {"bytes": len(stream), "sizes": [len(r) for r in recs], "sha256": hashlib.sha256(stream).hexdigest(),
                              "token_bytes": g["bytes"]}
    json.dump(out, open(os.path.join(GOLD, "defh.json"), "w"), indent=1)
    print("wrote defh.json:", out["enwik_like_300k"])


if __name__ == "__main__":
    main()
