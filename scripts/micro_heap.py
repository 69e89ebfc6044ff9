#!/usr/bin/env python3
"""k_huff_build alone, back to back (development aid): MI_CODEC_LIB=... python scripts/micro_heap.py [calls]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from compression_algorithms_amd import synth
from compression_algorithms_amd.huffman import HipShardEngine

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 200
eng = HipShardEngine()
ctx = eng.ctx
x = synth.enwik_like(10_000_000, seed=12345, device="cuda")
h, _ = eng.hist(x)
torch.cuda.synchronize()
d_info = torch.zeros(256, dtype=torch.uint8, device=ctx.device)
d_tree = torch.zeros(16384, dtype=torch.uint8, device=ctx.device)
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(calls):
        ctx.L.mi_huffman_build_dev(ctx.h, C.c_void_p(h.data_ptr()), C.c_void_p(d_info.data_ptr()), C.c_void_p(d_tree.data_ptr()), ctx.stream_ptr())
    torch.cuda.synchronize()
    print("k_huff_build back to back: %.1f us per call" % ((time.perf_counter() - t0) / calls * 1e6))
