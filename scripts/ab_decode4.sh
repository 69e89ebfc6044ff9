#!/bin/bash
# decoders: lib_base (HEAD before the change) against the working tree, same box
mkdir -p gpurun_out/r2
timeout -k 10 400 python -m pytest tests/test_decoders_gpu.py -q -m gpu > gpurun_out/r2/dec4_tests.txt 2>&1 || { tail -30 gpurun_out/r2/dec4_tests.txt; exit 1; }
tail -1 gpurun_out/r2/dec4_tests.txt
for lib in base new; do
  for wl in deflate-h deflate lz77w16; do
    if [ $lib = base ]; then export MI_CODEC_LIB=$PWD/compression_algorithms_amd/lib_base/libmi_codec.so; else unset MI_CODEC_LIB; fi
    timeout -k 10 200 python bench.py --workload $wl --bytes 1000000000 --steps 1 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r2/dec4_${lib}_${wl}.json 2>/dev/null || exit 1
    python -c "
import json; d=json.load(open('gpurun_out/r2/dec4_${lib}_${wl}.json')); print('$lib', '$wl', d['value'], 'decode', d.get('decode_gbps'), d['roundtrip'])"
  done
done
