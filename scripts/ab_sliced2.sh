#!/bin/bash
mkdir -p gpurun_out/r2
for wl in lz77w16-256k lz77w16-1m; do
  MI_LZS_SERIAL=1 python bench.py --workload $wl --bytes 100000000 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r2/ser_$wl.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r2/ser_$wl.json')); print('$wl serial', d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms_per_step'])"
done
python bench.py --workload lz77w16-1m --bytes 1000000000 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r2/big_1m.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r2/big_1m.json')); print('1m at 1e9 B', d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms_per_step'])"
python bench.py --workload lz77w16-256k --bytes 1000000000 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r2/big_256k.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r2/big_256k.json')); print('256k at 1e9 B', d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms_per_step'])"
