#!/usr/bin/env python3
"""Kernel timeline of one bench step from a rocprofv3 --kernel-trace csv: who runs beside whom (development aid).
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/x -- python3 bench.py ...;  python scripts/timeline.py gpurun_out/x"""
import csv
import glob
import sys

d = sys.argv[1]
files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        n = r.get("Kernel_Name") or r.get("kernel_name")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("(")[0][:40]))
rows.sort()
rows = [(s, e, n.replace("void ", "")) for s, e, n in rows]
ours = [r for r in rows if ("k_lz" in r[2] or "k_defh" in r[2]) and "decode" not in r[2]]
if not ours:
    sys.exit("no kernels")
# the last step: kernels after the last big gap
t_end = ours[-1][1]
# take the last 60 ms
import os
win = [r for r in ours if r[0] >= t_end - int(os.environ.get("TL_WINDOW_MS", "49")) * 1_000_000]
t0 = win[0][0]
print("kernels in window:", len(win))
for s, e, n in win:
    print(f"{(s - t0) / 1e6:9.3f} {(e - t0) / 1e6:9.3f} {(e - s) / 1e6:8.3f}  {n}")
# coverage: total time with >= 1 kernel of each group running
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
groups = {"A": ("k_lz2_partition", "k_lz2_find"), "B": ("k_lz2_mid", "k_lz2_big"), "C": ("k_lz_parse", "k_defh", "k_lz_concat", "k_lz_scan"),
          "lzs": ("k_lzs", "void k_lzs", "k_lzw")}
iv = [(s, e) for s, e, n in win]
print("any kernel busy", round(union(iv) / 1e6, 3), "ms of", round((win[-1][1] - t0) / 1e6, 3))
groups["fallback"] = ("k_lz_sort", "k_lz_emulate")
for g, pre in groups.items():
    iv = [(s, e) for s, e, n in win if any(n.startswith(p) for p in pre)]
    if iv:
        print(g, "busy", round(union(iv) / 1e6, 3), "ms of", round((win[-1][1] - t0) / 1e6, 3))
