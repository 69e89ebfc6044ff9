#!/bin/bash
# mode-H decoder: LUT bits / LDS diet variants, same box
mkdir -p gpurun_out/r2
timeout -k 10 300 python -m pytest tests/test_decoders_gpu.py tests/test_deflate_h_gpu.py -x -q -m gpu > gpurun_out/r2/lut_tests.txt 2>&1 || { tail -30 gpurun_out/r2/lut_tests.txt; exit 1; }
tail -1 gpurun_out/r2/lut_tests.txt
for lib in lib_base lib lib_l10 lib_l9 lib_base lib; do
  MI_CODEC_LIB=$PWD/compression_algorithms_amd/$lib/libmi_codec.so timeout -k 10 200 python bench.py --workload deflate-h --bytes 1000000000 --steps 1 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r2/lut_$lib.json 2>/dev/null || exit 1
  python -c "
import json; d=json.load(open('gpurun_out/r2/lut_$lib.json')); print('$lib', 'decode', d.get('decode_gbps'), d['roundtrip'])"
done
