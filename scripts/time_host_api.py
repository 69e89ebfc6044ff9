#!/usr/bin/env python3
"""PCIe-inclusive rate of the HOST-buffer entry points (what the drop-in compress() calls): pageable numpy buffers in,
stream out.  python scripts/time_host_api.py [bytes]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from compression_algorithms_amd import lz, synth, _lib
from compression_algorithms_amd.context import default_context

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
ctx = default_context()
x = synth.enwik_like(n, seed=12345).numpy()
ctx.L.mi_deflate_h_bound_bytes.restype = C.c_uint64
for name, p, fn, bound in [("deflate tokens (mi_lz_encode)", lz.params("deflate"), ctx.L.mi_lz_encode, lambda n_, p_: lz.bound_bytes(n_, p_)),
                           ("mode H (mi_deflate_h_encode)", lz.params("deflate"), ctx.L.mi_deflate_h_encode, lambda n_, p_: int(ctx.L.mi_deflate_h_bound_bytes(C.c_uint64(n_), C.byref(p_))))]:
    cap = int(bound(n, p)) + 64
    out = np.empty(cap, np.uint8)
    nblocks = (n + p.block - 1) // p.block
    bits = np.zeros(nblocks + 1, np.uint64)
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        st = fn(ctx.h, C.byref(p), C.c_void_p(x.ctypes.data), C.c_uint64(n), C.c_void_p(out.ctypes.data), C.c_uint64(cap), C.c_void_p(bits.ctypes.data))
        dt = time.perf_counter() - t0
        _lib.check(st, name)
        best = min(best, dt)
    print(f"{name}: {n / best / 1e9:.2f} GB/s ({best * 1e3:.1f} ms for {n} bytes -> {int(bits[-1]) // 8} bytes)")

# the way back: host stream in, host bytes out (what the drop-in decompress() calls)
for name, p, enc_fn, dec_fn in [("lz77 tokens (mi_lz_decode)", lz.params("lz77"), lz.compress, ctx.L.mi_lz_decode),
                                ("deflate tokens (mi_lz_decode)", lz.params("deflate"), lz.compress, ctx.L.mi_lz_decode),
                                ("mode H (mi_deflate_h_decode)", lz.params("deflate"), lz.compress_h, ctx.L.mi_deflate_h_decode)]:
    enc = enc_fn(x, p)
    bits = np.ascontiguousarray(enc.block_bits.cpu().numpy().astype(np.uint64))
    stream = np.frombuffer(enc.tobytes(), dtype=np.uint8).copy()
    del enc
    out = np.empty(n, np.uint8)
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        st = dec_fn(ctx.h, C.byref(p), C.c_void_p(stream.ctypes.data), C.c_uint64(len(stream)), C.c_void_p(bits.ctypes.data), C.c_void_p(out.ctypes.data), C.c_uint64(n))
        dt = time.perf_counter() - t0
        _lib.check(st, name)
        best = min(best, dt)
    assert np.array_equal(out, x)
    print(f"{name}: {n / best / 1e9:.2f} GB/s of output ({best * 1e3:.1f} ms, {len(stream)} stream bytes -> {n})")
