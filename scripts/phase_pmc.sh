#!/bin/bash
# per-phase instruction counts of k_lz2_find: the kernel leaves after phase k (MI_LZ_STOP_PHASE=k, measurement only — the
# output of such a run is wrong by construction), one counter pass per k; differences between successive k are the phases.
# Needs a MEASUREMENT build of the library (the shipped one ignores MI_LZ_STOP_PHASE):
#   make -C compression_algorithms_amd/csrc OUT=../lib_measure EXTRA=-DMI_MEASURE     (in the build container, before gpurun)
OUT=gpurun_out/phase_pmc; mkdir -p $OUT; export TMPDIR=/tmp
export MI_CODEC_LIB=$PWD/compression_algorithms_amd/lib_measure/libmi_codec.so
[ -f "$MI_CODEC_LIB" ] || { echo "build lib_measure first (see the header of this script)"; exit 1; }
cat > $OUT/drv.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from compression_algorithms_amd import lz, synth
from compression_algorithms_amd.context import Context
ctx = Context(0)
x = synth.enwik_like(67108864, seed=12345, device="cuda")
lz.compress(x, lz.params("deflate"), ctx)
torch.cuda.synchronize()
PY
for k in 1 2 3 4 5 6 7 0; do
  MI_LZ_STOP_PHASE=$k MI_LZ_NO_OVERLAP=1 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/k$k -- python3 $OUT/drv.py > $OUT/k$k.log 2>&1 || tail -3 $OUT/k$k.log
done
python3 - <<'PY'
import csv, glob
from collections import defaultdict
names = ["gather", "sort_home", "sweep", "sort_cluster", "permute", "lane_replay", "export+out"]
rows = {}
for k in (1, 2, 3, 4, 5, 6, 7, 0):
    acc = defaultdict(float)
    for f in glob.glob(f"gpurun_out/phase_pmc/k{k}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("k_lz2_find"):
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
    rows[k] = acc
prev = defaultdict(float)
print(f"{'phase':14s} {'VALU':>12s} {'SALU':>12s} {'LDS':>12s} {'VMEM_RD':>10s} {'VMEM_WR':>10s} {'WAVE_CYC':>14s} {'ACT_VALU':>12s}   (per launch of 1024 blocks, cumulative differences)")
for i, k in enumerate((1, 2, 3, 4, 5, 6, 7)):
    cur = rows[k]
    print(f"{names[i]:14s} " + " ".join(f"{(cur[c] - prev[c]) / 1e6:12.1f}" for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS")) +
          " " + " ".join(f"{(cur[c] - prev[c]) / 1e6:10.2f}" for c in ("SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR")) +
          f" {(cur['SQ_WAVE_CYCLES'] - prev['SQ_WAVE_CYCLES']) / 1e6:14.1f} {(cur['SQ_ACTIVE_INST_VALU'] - prev['SQ_ACTIVE_INST_VALU']) / 1e6:12.1f}")
    prev = cur
full = rows[0]
print("full kernel   " + " ".join(f"{full[c] / 1e6:12.1f}" for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS")))
PY
