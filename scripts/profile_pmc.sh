#!/bin/bash
# Runs on the GPU box (through gpurun): SQ counter passes behind the bound claims of DESIGN.md.
# Every group is its own rocprofv3 run (counters only: never combined with a trace domain); the program itself
# follows `--` (no env / bash -c hop).  usage: scripts/profile_pmc.sh <tag> [bench args...]
TAG=${1:-r02}; shift || true
OUT=gpurun_out/pmc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-extras --bytes 268435456 $*"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_GDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  echo "== pass $i: $grp"
  rocprofv3 --pmc $grp --output-format csv -d "$OUT/g$i" -- python3 bench.py $ARGS > "$OUT/bench_g$i.json" 2> "$OUT/g$i.log" || { echo "pass $i failed"; tail -3 "$OUT/g$i.log"; }
done
python3 scripts/summarise_pmc.py "$OUT" "$TAG"
rm -rf "$OUT"/g[0-9]*          # raw counter CSVs: tens of MiB per pass; the summary under gpurun_out/profiles/ is what is kept
