#!/usr/bin/env python3
"""Kernel times of the headline workload on a non-text input family (bench.py `adversarial`): where a cliff comes from.
    python scripts/adv_profile.py pages|runs|lowent|phrases|zeros [workload] [bytes]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from compression_algorithms_amd import synth
from compression_algorithms_amd.context import Context

kind = sys.argv[1] if len(sys.argv) > 1 else "pages"
wl = sys.argv[2] if len(sys.argv) > 2 else "deflate-h"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000_000
ctx = Context(0)
a = np.zeros(1 << 24, np.uint8) if kind == "zeros" else synth.family(kind, 4242, 1 << 24)
x = torch.from_numpy(a).cuda().repeat((n + len(a) - 1) // len(a))[:n].contiguous()
r = bench.measure(ctx, wl, x, 2, 1, None, False)
print(kind, wl, r["value"], "GB/s", r["ms_per_step"], "ms  ratio", r["ratio"], "rt", r["roundtrip"])
for k, v in sorted(r["roofline"]["all_kernels_ms_per_step"].items(), key=lambda kv: -kv[1]):
    print(f"   {k:28s} {v:10.3f} ms")
if os.environ.get("MI_LZ_DEBUG"):
    import ctypes as C
    out = (C.c_uint64 * 64)()
    ctx.L.mi_lz_debug_counters.argtypes = [C.c_void_p, C.c_void_p]
    ctx.L.mi_lz_debug_counters(ctx.h, out)
    v = list(out)
    print("dom: clusters seen %d (entries %d)  taken %d (entries %d)  setup cycles/cluster %.0f  replay cycles/entry %.1f  fast %d scan %d foreign %d bailed %d"
          % (v[48], v[49], v[50], v[51], v[52] / max(v[50], 1), v[53] / max(v[51], 1), v[54], v[55], v[56], v[57]))
    print('   entries placed k at a time', v[61], ' of them in the one-retirement-per-entry regime', v[62], ' serial retirements', v[63])
    if v[44]:
        print("tile kernel: %d tiles, cycles per tile: heads %.0f, load %.0f, lane replay %.0f, wave replay %.0f" % (v[44], v[40] / v[44], v[41] / v[44], v[42] / v[44], v[43] / v[44]))
        print("   slowest tile: lane replay %d cycles, wave replay %d cycles; largest tile %d entries" % (v[45], v[46], v[47]))
    print('   largest cluster', v[58], ' longest workgroup (ticks)', v[59], ' largest giant list', v[60])
