#!/usr/bin/env python3
"""Condenses the counter passes of scripts/profile_pmc.sh into one per-kernel table (profiles/<tag>_pmc.{json,txt}).
Counter values are summed over a kernel's dispatches and divided by their number: per-launch averages.  Units follow
MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_BUSY_CYCLES
counts per-SE busy cycles; FETCH_SIZE / WRITE_SIZE are KB (x1024 = bytes; FETCH_SIZE under-reports wide coalesced
reads by 2x on gfx950, so traffic = (2 x FETCH + WRITE) x 1024 is an upper estimate for gather-heavy kernels)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(out, "g*/**/*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        c = row["Counter_Name"]
        acc[k][c] += float(row["Counter_Value"])
        cnt[k][c] += 1
# rocprofv3 emits one row per (dispatch, counter[, dimension]); normalise by dispatches of the first counter of a kernel
summary = {}
for k in acc:
    d = {}
    for c in acc[k]:
        disp = None
        # dispatch count = number of distinct dispatches: approximate by rows / rows-per-dispatch (dimension instances)
        d[c] = acc[k][c]
    summary[k] = d
# dispatch counts from a dedicated scan (Dispatch_Id distinct per kernel)
disp = defaultdict(set)
for f in glob.glob(os.path.join(out, "g1/**/*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        disp[row["Kernel_Name"].split("(")[0]].add(row.get("Dispatch_Id"))
table = {}
for k, d in summary.items():
    n = max(len(disp.get(k, ())), 1)
    e = {c: v / n for c, v in d.items()}
    e["launches"] = n
    wc = e.get("SQ_WAVE_CYCLES")
    if wc:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU",
                  "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM"):
            if c in e:
                e[c + "/WAVE_CYCLES"] = round(e[c] / wc, 4)
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        e["hbm_bytes_fetch_x2"] = (2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024
        e["hbm_bytes_raw"] = (e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024
    table[k] = e
os.makedirs("gpurun_out/profiles", exist_ok=True)
json.dump(table, open(f"gpurun_out/profiles/{tag}_pmc.json", "w"), indent=1)
cols = ["launches", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM",
        "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAIT_ANY/WAVE_CYCLES", "SQ_WAIT_INST_ANY/WAVE_CYCLES", "SQ_ACTIVE_INST_ANY/WAVE_CYCLES",
        "SQ_WAIT_INST_LDS/WAVE_CYCLES", "SQ_ACTIVE_INST_VALU/WAVE_CYCLES", "SQ_ACTIVE_INST_SCA/WAVE_CYCLES", "SQ_ACTIVE_INST_LDS/WAVE_CYCLES",
        "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "hbm_bytes_fetch_x2"]
with open(f"gpurun_out/profiles/{tag}_pmc.txt", "w") as fh:
    fh.write(f"rocprofv3 --pmc passes (one group per run), per-launch averages: {tag}\n")
    for k, e in sorted(table.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        if not k.startswith("k_"):
            continue
        fh.write(f"\n{k}\n")
        for c in cols:
            if c in e:
                v = e[c]
                fh.write(f"  {c:36s} {v:18.4f}\n" if isinstance(v, float) and v < 10 else f"  {c:36s} {v:18.0f}\n")
print(open(f"gpurun_out/profiles/{tag}_pmc.txt").read())
