#!/bin/bash
mkdir -p gpurun_out/r2
for wl in deflate-h deflate lz77w16 lz77w14; do
  python bench.py --workload $wl --bytes 1000000000 --steps 1 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r2/dec2_${wl}.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r2/dec2_${wl}.json')); print('$wl', d['value'], 'decode', d.get('decode_gbps'), d['roundtrip'])"
done
