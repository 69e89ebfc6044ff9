#!/bin/bash
mkdir -p gpurun_out/r2
B="python bench.py --workload deflate-h --steps 5 --warmup 1 --no-cpu-baseline --no-extras"
for lib in lib_base lib lib_base lib; do
  MI_CODEC_LIB=$PWD/compression_algorithms_amd/$lib/libmi_codec.so $B > gpurun_out/r2/abh_$lib.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r2/abh_$lib.json')); print('$lib', d['value'], d['ms_per_step'], d['roundtrip'], d['roofline']['all_kernels_ms_per_step'].get('k_lz_parse_emit'))"
done
