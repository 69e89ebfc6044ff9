#!/bin/bash
# decoder ring size A/B: MI_LZ_DECODE_RING=65536 (the ring is the window, as before) vs the default 16 KiB ring with far reads
mkdir -p gpurun_out/r2
for wl in deflate-h deflate lz77w16 lz77w14; do
  for ring in 8192 4096; do
    nb=1000000000; [ $wl = lz77w16-256k ] && nb=100000000
    MI_LZ_DECODE_RING=$ring python bench.py --workload $wl --bytes $nb --steps 1 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r2/dec3_${wl}_$ring.json 2>/dev/null
    python -c "
import json; d=json.load(open('gpurun_out/r2/dec3_${wl}_$ring.json')); print('$wl ring $ring decode', d.get('decode_gbps'), d['roundtrip'])"
  done
done
