#!/bin/bash
mkdir -p gpurun_out/r2
timeout -k 10 900 python -m pytest tests/test_lz_encode_gpu.py tests/test_deflate_h_gpu.py tests/test_corrupt_gpu.py tests/test_lz_wide_gpu.py tests/test_dropin.py tests/test_bounds.py tests/test_fuzz_gpu.py -x -q -m gpu 2>&1 | tail -5
for wl in deflate-h deflate lz77w16 lz77w14; do
  for lib in lib_base lib; do
    MI_CODEC_LIB=$PWD/compression_algorithms_amd/$lib/libmi_codec.so python bench.py --workload $wl --bytes 1000000000 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r2/dec_${wl}_$lib.json 2>gpurun_out/r2/dec_${wl}_$lib.err
    python - <<PY
import json
try:
    d=json.load(open('gpurun_out/r2/dec_${wl}_$lib.json')); print('$wl $lib', d['value'], 'decode', d.get('decode_gbps'), d['roundtrip'])
except Exception as e: print('$wl $lib failed', e)
PY
  done
done
for wl in lz77w16-256k lz77w16-1m; do
    python bench.py --workload $wl --bytes 100000000 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r2/dec_${wl}.json 2>gpurun_out/r2/dec_${wl}.err
    python -c "
import json; d=json.load(open('gpurun_out/r2/dec_${wl}.json')); print('$wl', d['value'], 'decode', d.get('decode_gbps'), d['roundtrip'])"
done
