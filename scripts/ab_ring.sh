#!/bin/bash
mkdir -p gpurun_out/r2
for ring in 2048 4096 8192; do
  for wl in deflate-h deflate lz77w16; do
    MI_LZ_DECODE_RING=$ring timeout -k 10 200 python bench.py --workload $wl --bytes 1000000000 --steps 1 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r2/ring_${ring}_${wl}.json 2>/dev/null || exit 1
    python -c "
import json; d=json.load(open('gpurun_out/r2/ring_${ring}_${wl}.json')); print('ring $ring', '$wl', 'decode', d.get('decode_gbps'), d['roundtrip'])"
  done
done
