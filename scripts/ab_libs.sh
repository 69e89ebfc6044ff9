#!/bin/bash
# same-box A/B of several builds of the same ABI: scripts/ab_libs.sh <tag> <workload> <libdir>...
TAG=$1; WL=$2; shift; shift
mkdir -p gpurun_out/r2
B="python bench.py --workload $WL --steps 5 --warmup 1 --no-cpu-baseline --no-extras"
for rep in 1 2; do
for L in "$@"; do
  name=$(basename $L)_$rep
  MI_CODEC_LIB=$PWD/$L/libmi_codec.so $B > gpurun_out/r2/${TAG}_$name.json 2> gpurun_out/r2/${TAG}_$name.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r2/${TAG}_$name.json"))
print("$name", d["value"], d["ms_per_step"], d["roundtrip"], {k: v for k, v in d["roofline"]["all_kernels_ms_per_step"].items()})
PY
done
done
for L in "$@"; do
  name=$(basename $L)_serial
  MI_LZ_NO_OVERLAP=1 MI_CODEC_LIB=$PWD/$L/libmi_codec.so $B > gpurun_out/r2/${TAG}_$name.json 2> gpurun_out/r2/${TAG}_$name.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r2/${TAG}_$name.json"))
print("$name", d["value"], d["ms_per_step"], d["roundtrip"], {k: v for k, v in d["roofline"]["all_kernels_ms_per_step"].items()})
PY
done
