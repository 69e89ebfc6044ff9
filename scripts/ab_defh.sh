#!/bin/bash
mkdir -p gpurun_out/r2
B="python bench.py --workload deflate-h --steps 5 --warmup 1 --no-cpu-baseline --no-extras"
for lib in lib lib_ab256 lib_ab128 lib lib_ab256 lib_ab128; do
  MI_CODEC_LIB=$PWD/compression_algorithms_amd/$lib/libmi_codec.so $B > gpurun_out/r2/abd_$lib.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r2/abd_$lib.json')); print('$lib', d['value'], d['ms_per_step'], d['roundtrip'], d['roofline']['all_kernels_ms_per_step'].get('k_defh_encode'))"
done
