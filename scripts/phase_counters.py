#!/usr/bin/env python3
"""In-kernel phase shares of k_lz2_find / radix_pass / k_lz_parse_emit (clock64 counters, MI_LZ_DEBUG=1; development aid).
    MI_LZ_DEBUG=1 python scripts/phase_counters.py [bytes]"""
import ctypes as C
import os
import sys

os.environ.setdefault("MI_LZ_DEBUG", "1")
os.environ.setdefault("MI_LZ_NO_OVERLAP", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from compression_algorithms_amd import lz, synth
from compression_algorithms_amd.context import Context

n = int(sys.argv[1]) if len(sys.argv) > 1 else 268435456
ctx = Context(0)
x = synth.enwik_like(n, seed=12345, device="cuda")
p = lz.params("deflate")
(lz.compress_h if os.environ.get("MI_PHASE_MODE_H") else lz.compress)(x, p, ctx)
torch.cuda.synchronize()
out = (C.c_uint64 * 64)()
ctx.L.mi_lz_debug_counters.argtypes = [C.c_void_p, C.c_void_p]
ok = ctx.L.mi_lz_debug_counters(ctx.h, out)
v = np.array(list(out), dtype=np.float64)
names = ["gather", "sort_home", "sweep", "sort_cluster", "permute", "lane_replay", "export+out"]
tot = v[:7].sum() + v[11:14].sum()
v[5] += v[11:14].sum()
print("k_lz2_find parts:", int(v[15]), " cycles/part:", int(tot / max(v[15], 1)))
for k, nm in enumerate(names):
    print(f"  {nm:14s} {100 * v[k] / tot:5.1f} %   {v[k] / max(v[15], 1):9.0f} cycles/part")
np_ = max(v[15], 1)
print("  lane_replay split: classify %.0f  bucket %.0f  reserve+quiet %.0f  (replay loop = rest) cycles/part" % (v[11] / np_, v[12] / np_, v[13] / np_))
print("  per part: entries %.0f  clusters %.0f  lane-replayed clusters (2..7) %.0f  exported clusters %.0f  exported entries (padded) %.0f  quiet clusters >= 8: %.0f"
      % (v[27] / np_, v[28] / np_, v[23] / np_, v[24] / np_, v[25] / np_, v[26] / np_))
rp = v[8:11]
print("radix_pass (instrumented passes only): count %.0f  offsets %.0f  scatter %.0f cycles/part" % tuple(rp / max(v[15], 1)))
pe = v[16:23]
pt = pe.sum()
print("k_lz_parse_emit blocks:", int(v[31]), " cycles/block:", int(pt / max(v[31], 1)))
for k, nm in enumerate(["load", "lengths", "exit tables", "compose", "chunks", "scan+put", "emit"]):
    print(f"  {nm:14s} {100 * pe[k] / pt:5.1f} %   {pe[k] / max(v[31], 1):9.0f} cycles/block")

pp = v[32:39]
nb_ = max(v[47], 1)
if v[47]:
    print("k_lz2_partition blocks:", int(v[47]), " cycles/block:", int(pp.sum() / nb_))
    for k, nm in enumerate(["load", "hash+count", "certificate+prefix", "cuts", "part table", "part of every position", "radix pass"]):
        print(f"  {nm:24s} {100 * pp[k] / pp.sum():5.1f} %   {pp[k] / nb_:9.0f} cycles/block")
