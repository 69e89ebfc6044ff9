#!/usr/bin/env python3
"""A longer randomized parity campaign than tests/test_fuzz_gpu.py (development aid; run on an MI355X):
    python scripts/fuzz_campaign.py [cases] [first_seed] [wide]
"wide": every case is the lz77 flavour on blocks of 128 KiB .. 1 MiB (the time-sliced finder, its fallback on runs/pages).
Every case: a random input family / length / block size / flavour; find() at every position against the oracle's literal
table, the stream round trip, and for deflate the token stream against the oracle.  Prints one line per failure."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_fuzz_gpu as F
from compression_algorithms_amd import lz
from oracle import orc

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
wide = len(sys.argv) > 3 and sys.argv[3] == "wide"
kinds = ["text", "lowent", "phrases", "runs", "pages"]
fails = 0
t0 = time.time()
for i in range(cases):
    seed = seed0 + i
    rng = np.random.default_rng(seed)
    kind = kinds[int(rng.integers(0, len(kinds)))]
    flavour, wbits = [("deflate", None), ("lz77", 14), ("lz77", 16)][int(rng.integers(0, 3))]
    block = [65536, 65536, 65536, 20000, 4096, 50000][int(rng.integers(0, 6))]
    n = int(rng.integers(1, 5)) * block + int(rng.integers(1, block))
    if i % 10 == 9 and flavour == "lz77":
        block = [131072, 262144][int(rng.integers(0, 2))]
        n = int(rng.integers(block, 3 * block))
    if wide:
        flavour, wbits = "lz77", [14, 16, 16][int(rng.integers(0, 3))]
        block = [131072, 262144, 524288, 1048576, 196608][int(rng.integers(0, 5))]
        n = int(rng.integers(block // 2, int(2.2 * block)))
    data = F._family(kind, seed, n)
    p = lz.params(flavour, wbits, block)
    try:
        if block > 65536:
            got = lz.find_all32(data, p).cpu().numpy().view(np.uint32)
        else:
            got = lz.find_all(data, p).cpu().numpy().view(np.uint16)
        for at in range(0, n, block):
            want = orc.find_all(data[at:at + block], p.wbits, p.tbits, bool(p.deflate))
            if block <= 65536:
                want = np.where(want == 0xFFFFFFFF, 0xFFFF, want).astype(np.uint16)
            g = got[at:at + block]
            if not np.array_equal(g, want):
                bad = np.flatnonzero(g != want)
                raise AssertionError(f"find: {bad.size} mismatches from {at + bad[0]}")
        st = lz.compress(data, p)
        if not np.array_equal(lz.decompress(st).cpu().numpy(), data):
            raise AssertionError("round trip")
        if flavour == "deflate":
            tok, _ = orc.deflate_stream(data, block, True)
            if st.tobytes() != tok.tobytes():
                raise AssertionError("token stream")
            sth = lz.compress_h(data, p)
            if not np.array_equal(lz.decompress_h(sth).cpu().numpy(), data):
                raise AssertionError("mode H round trip")
    except Exception as e:
        fails += 1
        print(f"FAIL seed {seed} {kind} n={n} block={block} {flavour} w{wbits}: {e}", flush=True)
    if i % 20 == 19:
        print(f"{i + 1} cases, {fails} failures, {time.time() - t0:.0f} s", flush=True)
print(f"done: {cases} cases, {fails} failures")
sys.exit(1 if fails else 0)
