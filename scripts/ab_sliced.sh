#!/bin/bash
# same-box A/B of the two finders for lz77 blocks above 64 KiB: MI_LZW_SLICED=0 (lzw.hip, whole-block clusters) vs the default (lzs.hip)
mkdir -p gpurun_out/r2
for wl in lz77w16-256k lz77w16-1m; do
  for sl in 0 1; do
    MI_LZW_SLICED=$sl python bench.py --workload $wl --bytes 100000000 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r2/sl_${wl}_$sl.json 2>gpurun_out/r2/sl_${wl}_$sl.err
    python - <<PY
import json
d=json.load(open('gpurun_out/r2/sl_${wl}_$sl.json'))
print('$wl sliced=$sl', d['value'], 'GB/s', d['ms_per_step'], 'ms; decode', d.get('decode_gbps'), d['roundtrip'])
print('   ', d['roofline'].get('all_kernels_ms_per_step'))
PY
  done
done
for g in 1; do
  MI_LZS_GROUPS=$g python bench.py --workload lz77w16-1m --bytes 100000000 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r2/sl_g$g.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r2/sl_g$g.json')); print('1m groups=$g', d['value'], d['ms_per_step'], d['roundtrip'])"
done
