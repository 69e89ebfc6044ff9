#!/bin/bash
# round-2 A/B: persistent grids
mkdir -p gpurun_out/r2
B="python bench.py --workload deflate --steps 5 --warmup 1 --no-cpu-baseline --no-extras"
run() { name=$1; shift; env "$@" $B > gpurun_out/r2/ab_$name.json 2> gpurun_out/r2/ab_$name.err; python - <<PY
import json
d=json.load(open("gpurun_out/r2/ab_$name.json"))
print("$name", d["value"], d["ms_per_step"], {k: v for k, v in d["roofline"]["all_kernels_ms_per_step"].items()})
PY
}
run base X=1
run find384 MI_LZ_FIND_WGS=384
run find448 MI_LZ_FIND_WGS=448
run find1024 MI_LZ_FIND_WGS=1024
run nofb MI_LZ_UNSAFE_NO_FALLBACK=1
run waves2 MI_LZ_REPLAY_WAVES=4
B="python bench.py --workload deflate --bytes 125000000 --steps 10 --warmup 2 --no-cpu-baseline --no-extras"
run s125_base X=1
run s125_b256 MI_LZ_BATCH=256
run s125_b320 MI_LZ_BATCH=320
run s125_b480 MI_LZ_BATCH=480
run s125_b640 MI_LZ_BATCH=640
