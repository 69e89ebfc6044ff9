#!/bin/bash
# mode H, lib_base against the working tree, same box; a parity subset first
mkdir -p gpurun_out/r2
timeout -k 10 400 python -m pytest tests/test_lz_find_gpu.py tests/test_lz_encode_gpu.py -x -q -m gpu > gpurun_out/r2/abh2_tests.txt 2>&1 || { tail -30 gpurun_out/r2/abh2_tests.txt; exit 1; }
tail -1 gpurun_out/r2/abh2_tests.txt
B="python bench.py --workload deflate-h --steps 5 --warmup 1 --no-cpu-baseline --no-extras"
for lib in lib_base lib lib_base lib; do
  MI_CODEC_LIB=$PWD/compression_algorithms_amd/$lib/libmi_codec.so timeout -k 10 200 $B > gpurun_out/r2/abh_$lib.json 2>/dev/null || exit 1
  python -c "
import json; d=json.load(open('gpurun_out/r2/abh_$lib.json')); k=d['roofline']['all_kernels_ms_per_step']; print('$lib', d['value'], d['ms_per_step'], d['roundtrip'], 'find', k.get('k_lz2_find'), 'parse', k.get('k_lz_parse_emit'))"
done
