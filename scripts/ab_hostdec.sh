#!/bin/bash
# host-buffer decoders: one shot against chunks
mkdir -p gpurun_out/r2
timeout -k 10 300 python -m pytest tests/test_dropin.py -q -m gpu -k "in_chunks" > gpurun_out/r2/hostdec_tests.txt 2>&1 || { tail -30 gpurun_out/r2/hostdec_tests.txt; exit 1; }
tail -2 gpurun_out/r2/hostdec_tests.txt
for cfg in "100000000 1" "2048 3" "2048 4" "4096 2" "1024 4" "4096 3"; do
  set -- $cfg
  echo "== MI_HOST_DECODE_CHUNK_BLOCKS=$1 AHEAD=$2"
  MI_HOST_DECODE_CHUNK_BLOCKS=$1 MI_HOST_DECODE_AHEAD=$2 timeout -k 10 200 python scripts/time_host_api.py 2>&1 | grep -i "decode" || exit 1
done
