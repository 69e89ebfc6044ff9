#!/usr/bin/env python3
"""Data preparation for the reference's drivers — the offline half of /root/reference/get_data.sh.

get_data.sh:3-5 downloads enwik9 (no network here); lines 6-8 derive the smaller corpora:
    head -c 100000000 enwik9 > enwik8;  head -c 10000000 enwik8 > enwik7;  head -c 1000000 enwik7 > enwik6
This helper does the same derivations from an enwik9 the box already holds (--source), or, when there is none, from the
seeded synthetic enwik-shaped generator (compression_algorithms_amd/synth.py) — and says which.
The drivers open ../../data/<name> relative to algorithms/<dir>/, so --dir is normally <tree>/data.

    python scripts/prep_data.py --dir data [--source /path/to/enwik9] [--size 1000000000] [--device cuda:0]
"""
import argparse
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DERIVED = (("enwik8", 100_000_000), ("enwik7", 10_000_000), ("enwik6", 1_000_000))


def head_c(src, dst, nbytes, chunk=1 << 24):
    """head -c nbytes src > dst"""
    left = nbytes
    with open(src, "rb") as fi, open(dst, "wb") as fo:
        while left > 0:
            b = fi.read(min(chunk, left))
            if not b:
                break
            fo.write(b)
            left -= len(b)
    return nbytes - left


def prepare(out_dir, source=None, size=1_000_000_000, device="cpu", seed=12345):
    os.makedirs(out_dir, exist_ok=True)
    e9 = os.path.join(out_dir, "enwik9")
    if source:
        if os.path.abspath(source) != os.path.abspath(e9):
            shutil.copyfile(source, e9)
        kind = f"copy of {source}"
    else:
        from compression_algorithms_amd import synth
        left, at = size, 0
        with open(e9, "wb") as fo:          # generated in 256 MB pieces (one seed per piece) to bound memory
            while left > 0:
                n = min(left, 1 << 28)
                fo.write(synth.enwik_like(n, seed=seed + at, device=device).cpu().numpy().tobytes())
                left -= n
                at += 1
        kind = f"synthetic enwik-shaped, seed {seed}, {size} bytes"
    made = {"enwik9": os.path.getsize(e9)}
    prev = e9
    for name, nbytes in DERIVED:                               # get_data.sh:6-8
        dst = os.path.join(out_dir, name)
        made[name] = head_c(prev, dst, nbytes)
        prev = dst
    return kind, made


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", default="data")
    ap.add_argument("--source", default=None, help="an existing enwik9 (real data); default: synthetic")
    ap.add_argument("--size", type=int, default=1_000_000_000, help="bytes of the synthetic enwik9")
    ap.add_argument("--device", default="cpu")
    a = ap.parse_args()
    kind, made = prepare(a.dir, a.source, a.size, a.device)
    print(f"{a.dir}: enwik9 = {kind}")
    for k, v in made.items():
        print(f"  {k}: {v} bytes")
