#!/bin/bash
mkdir -p gpurun_out/r2
B="python bench.py --workload deflate-h --bytes 125000000 --steps 20 --warmup 3 --no-cpu-baseline --no-extras"
for b in 1024 954 640 480 384 320 256; do
  MI_LZ_BATCH=$b $B > gpurun_out/r2/batch_$b.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r2/batch_$b.json')); print('batch $b', d['value'], d['ms_per_step'])"
done
