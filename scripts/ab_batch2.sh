#!/bin/bash
mkdir -p gpurun_out/r2
B="python bench.py --workload deflate-h --steps 5 --warmup 1 --no-cpu-baseline --no-extras"
for b in 1024 1536 2048 3072 4096 1024; do
  MI_LZ_BATCH=$b $B > gpurun_out/r2/batch2_$b.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r2/batch2_$b.json')); print('batch $b', d['value'], d['ms_per_step'], d['roundtrip'])"
done
MI_LZ_BATCH=2048 python bench.py --workload deflate-h --bytes 125000000 --steps 20 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r2/batch2_s125.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r2/batch2_s125.json')); print('125MB one batch of 1908', d['value'], d['ms_per_step'], d['roundtrip'])"
