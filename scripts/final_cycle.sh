#!/bin/bash
# full GPU suite, smoke, default bench line -> gpurun_out/r2/
mkdir -p gpurun_out/r2
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r2/final_gpu_tests.txt 2>&1 || { tail -40 gpurun_out/r2/final_gpu_tests.txt; exit 1; }
tail -1 gpurun_out/r2/final_gpu_tests.txt
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 || exit 1
timeout -k 10 400 python bench.py > gpurun_out/r2/bench_default_final12.json 2> gpurun_out/r2/bench_default_final12.err || { tail -20 gpurun_out/r2/bench_default_final12.err; exit 1; }
cat gpurun_out/r2/bench_default_final12.json
