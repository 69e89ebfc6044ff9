#!/bin/bash
# same-box A/B: compression_algorithms_amd/lib_base (built from the last commit) against the working tree's library
# usage: scripts/ab_same_box.sh <tag> [workload]   (extra env pairs: NAME=VALUE ... after the workload, applied to the "new" runs)
TAG=${1:-ab}; WL=${2:-deflate}; shift; shift
mkdir -p gpurun_out/r2
B="python bench.py --workload $WL --steps 5 --warmup 1 --no-cpu-baseline --no-extras"
run() { name=$1; shift; env "$@" $B > gpurun_out/r2/${TAG}_$name.json 2> gpurun_out/r2/${TAG}_$name.err; python - <<PY
import json
d=json.load(open("gpurun_out/r2/${TAG}_$name.json"))
print("$name", d["value"], d["ms_per_step"], d["roundtrip"], {k: v for k, v in d["roofline"]["all_kernels_ms_per_step"].items()})
PY
}
BASE=$PWD/compression_algorithms_amd/lib_base/libmi_codec.so
run old MI_CODEC_LIB=$BASE
run new X=1 "$@"
run old_b MI_CODEC_LIB=$BASE
run new_b X=1 "$@"
run old_serial MI_CODEC_LIB=$BASE MI_LZ_NO_OVERLAP=1
run new_serial MI_LZ_NO_OVERLAP=1 "$@"
