#!/bin/bash
mkdir -p gpurun_out/r2
for wl in lz77w16 lz77w14; do
  for lib in lib_base lib lib_base lib; do
    nb=1000000000; [ $wl != deflate ] && nb=100000000
    MI_CODEC_LIB=$PWD/compression_algorithms_amd/$lib/libmi_codec.so python bench.py --workload $wl --bytes $nb --steps 5 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r2/abt_$lib.json 2>/dev/null
    python -c "
import json; d=json.load(open('gpurun_out/r2/abt_$lib.json')); print('$wl $lib', d['value'], d['ms_per_step'], d['roundtrip'])"
  done
done
