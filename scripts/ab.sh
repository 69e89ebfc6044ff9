#!/bin/bash
# Same-box A/B of builds and/or environment switches through bench.py (one script for what used to be 22 ab_*.sh):
#
#   scripts/ab.sh TAG WORKLOAD VARIANT [VARIANT ...]
#     VARIANT = name[:lib=<dir under compression_algorithms_amd/>][:ENV=VALUE]...
#       scripts/ab.sh lds deflate-h base:lib=lib_base new
#       scripts/ab.sh ring deflate  w:MI_LZ_DECODE_RING=65536 r8k:MI_LZ_DECODE_RING=8192
#       AB_BYTES=125000000 AB_STEPS=20 scripts/ab.sh batch deflate-h b1024:MI_LZ_BATCH=1024 b640:MI_LZ_BATCH=640
#   AB_BYTES (1e9)  AB_STEPS (5)  AB_WARMUP (1)  AB_REPS (2: variants alternate, so drift shows)  AB_OUT (gpurun_out/r3)
#   AB_TESTS="tests/test_lz_find_gpu.py tests/test_lz_encode_gpu.py"   parity subset that must pass first
# Keep a copy of the library before a change with:  cp -r compression_algorithms_amd/lib compression_algorithms_amd/lib_base
set -u
TAG=$1; WL=$2; shift; shift
OUT=${AB_OUT:-gpurun_out/r3}; mkdir -p "$OUT"
BYTES=${AB_BYTES:-1000000000}; STEPS=${AB_STEPS:-5}; WARM=${AB_WARMUP:-1}; REPS=${AB_REPS:-2}
if [ -n "${AB_TESTS:-}" ]; then
  timeout -k 10 900 python -m pytest $AB_TESTS -x -q -m gpu > "$OUT/${TAG}_tests.txt" 2>&1 || { tail -30 "$OUT/${TAG}_tests.txt"; exit 1; }
  tail -1 "$OUT/${TAG}_tests.txt"
fi
for rep in $(seq 1 "$REPS"); do
  for V in "$@"; do
    IFS=: read -r -a F <<< "$V"
    name=${F[0]}; ENVS=()
    for kv in "${F[@]:1}"; do
      if [[ $kv == lib=* ]]; then ENVS+=("MI_CODEC_LIB=$PWD/compression_algorithms_amd/${kv#lib=}/libmi_codec.so"); else ENVS+=("$kv"); fi
    done
    J="$OUT/${TAG}_${name}_$rep.json"
    # (env before python: nothing has touched the GPU yet in this shell)
    env "${ENVS[@]}" timeout -k 10 600 python bench.py --workload "$WL" --bytes "$BYTES" --steps "$STEPS" --warmup "$WARM" \
        --no-cpu-baseline --no-extras > "$J" 2> "${J%.json}.err" || { echo "$name failed"; tail -5 "${J%.json}.err"; exit 1; }
    python - "$J" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = (d.get("roofline") or {}).get("all_kernels_ms_per_step", {})
print(f"{sys.argv[2]:>12}  {d['value']:8.3f} GB/s  {d['ms_per_step']:9.3f} ms  rt={d['roundtrip']}  decode={d.get('decode_gbps')}  bytes={d['config']['compressed_bytes_job']}  " +
      " ".join(f"{a.replace('k_', '')}={b}" for a, b in k.items()))
PY
  done
done
