#!/bin/bash
mkdir -p gpurun_out/r2
timeout -k 10 300 python -m pytest tests/test_dropin.py -q -m gpu -k "in_chunks" > gpurun_out/r2/hostenc_tests.txt 2>&1 || { tail -30 gpurun_out/r2/hostenc_tests.txt; exit 1; }
tail -1 gpurun_out/r2/hostenc_tests.txt
for lib in base new base new; do
  if [ $lib = base ]; then export MI_CODEC_LIB=$PWD/compression_algorithms_amd/lib_base/libmi_codec.so; else unset MI_CODEC_LIB; fi
  echo "== $lib"
  timeout -k 10 200 python scripts/time_host_api.py 2>&1 | grep -i "encode" || exit 1
done
