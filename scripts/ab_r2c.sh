#!/bin/bash
mkdir -p gpurun_out/r2
B="python bench.py --workload deflate --steps 5 --warmup 1 --no-cpu-baseline --no-extras"
run() { name=$1; shift; env "$@" $B > gpurun_out/r2/ab_$name.json 2> gpurun_out/r2/ab_$name.err; python - <<PY
import json
d=json.load(open("gpurun_out/r2/ab_$name.json"))
print("$name", d["value"], d["ms_per_step"], {k: v for k, v in d["roofline"]["all_kernels_ms_per_step"].items()})
PY
}
BASE=$PWD/compression_algorithms_amd/lib_base/libmi_codec.so
run old MI_CODEC_LIB=$BASE
run f0_r0 MI_LZ_FIND_WGS=0 MI_LZ_REPLAY_WAVES=0
run f0_rdef MI_LZ_FIND_WGS=0
run fdef_r0 MI_LZ_REPLAY_WAVES=0
run f22k_r0 MI_LZ_FIND_WGS=22528 MI_LZ_REPLAY_WAVES=0
run f0_r64 MI_LZ_FIND_WGS=0 MI_LZ_REPLAY_WAVES=64
run f0_r256 MI_LZ_FIND_WGS=0 MI_LZ_REPLAY_WAVES=256
run old_serial MI_CODEC_LIB=$BASE MI_LZ_NO_OVERLAP=1
run new_serial MI_LZ_NO_OVERLAP=1
run new_serial_f0r0 MI_LZ_NO_OVERLAP=1 MI_LZ_FIND_WGS=0 MI_LZ_REPLAY_WAVES=0
