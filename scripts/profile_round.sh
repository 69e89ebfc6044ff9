#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace statistics and HBM traffic counters of the
# bench command.  Counters are collected in their own passes (never with a trace domain).
# usage: scripts/profile_round.sh <tag> [bench args...]
set -e
TAG=${1:-r01}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-extras $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py $ARGS > "$OUT/bench_trace.json" 2> "$OUT/trace.log" || { tail -5 "$OUT/trace.log"; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py $ARGS > "$OUT/bench_fetch.json" 2> "$OUT/fetch.log" || { tail -5 "$OUT/fetch.log"; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 bench.py $ARGS > "$OUT/bench_write.json" 2> "$OUT/write.log" || { tail -5 "$OUT/write.log"; exit 1; }
python3 scripts/summarise_profile.py "$OUT" "$TAG"
