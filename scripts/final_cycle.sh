#!/bin/bash
# full GPU suite, smoke, default bench line, decoder LUT variants, profile refresh -> gpurun_out/
mkdir -p gpurun_out/r2 gpurun_out/profiles
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/r2/final_gpu_tests.txt 2>&1 || { tail -40 gpurun_out/r2/final_gpu_tests.txt; exit 1; }
tail -1 gpurun_out/r2/final_gpu_tests.txt
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 || exit 1
timeout -k 10 300 python bench.py > gpurun_out/r2/bench_default_final13.json 2> gpurun_out/r2/bench_default_final13.err || { tail -20 gpurun_out/r2/bench_default_final13.err; exit 1; }
cat gpurun_out/r2/bench_default_final13.json
for cfg in "lib_l11 4096" "lib 4096" "lib_l8 4096" "lib_l7 4096" "lib 8192" "lib_l8 8192"; do
  set -- $cfg
  MI_LZ_DECODE_RING=$2 MI_CODEC_LIB=$PWD/compression_algorithms_amd/$1/libmi_codec.so timeout -k 10 100 python bench.py --workload deflate-h --bytes 1000000000 --steps 1 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r2/lut2.json 2>/dev/null || exit 1
  python -c "
import json; d=json.load(open('gpurun_out/r2/lut2.json')); print('$1 ring $2', 'decode', d.get('decode_gbps'), d['roundtrip'])"
done
cd /tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 300 bash scripts/profile_round.sh r02c_deflate-h --workload deflate-h > gpurun_out/r2/prof_r02c.txt 2>&1 || { tail -5 gpurun_out/r2/prof_r02c.txt; }
rm -rf gpurun_out/prof_r02c_deflate-h/trace gpurun_out/prof_r02c_deflate-h/pmc_fetch gpurun_out/prof_r02c_deflate-h/pmc_write
tail -25 gpurun_out/r2/prof_r02c.txt
