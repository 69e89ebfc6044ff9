#!/bin/bash
# round-2 A/B on ONE box: baseline library (lib_base, built from the last commit) against the working tree
mkdir -p gpurun_out/r2
B="python bench.py --workload deflate --steps 5 --warmup 1 --no-cpu-baseline --no-extras"
run() { name=$1; shift; env "$@" $B > gpurun_out/r2/ab_$name.json 2> gpurun_out/r2/ab_$name.err; python - <<PY
import json
d=json.load(open("gpurun_out/r2/ab_$name.json"))
print("$name", d["value"], d["ms_per_step"], {k: v for k, v in d["roofline"]["all_kernels_ms_per_step"].items()})
PY
}
BASE=$PWD/compression_algorithms_amd/lib_base/libmi_codec.so
run old1 MI_CODEC_LIB=$BASE
run new1 X=1
run old2 MI_CODEC_LIB=$BASE
run new2 X=1
run new_waves8 MI_LZ_REPLAY_WAVES=8
run new_find320 MI_LZ_FIND_WGS=320
