#!/bin/bash
# Runs the GPU steps of one gpurun call, one after the other, each under its own `timeout -k`, output under gpurun_out/<dir>/.
# A step that fails (tests red) does not stop the next one; a step that is KILLED at its limit, or that ends on a signal (a GPU
# fault), does — nothing may be started on a box whose GPU may be wedged.
#   usage: scripts/gpu_steps.sh <outdir> <<'STEPS'
#          name|seconds|command ...
#          STEPS
out="gpurun_out/$1"; mkdir -p "$out"
cd /tmp 2>/dev/null && export TMPDIR=/tmp; cd - >/dev/null
while IFS='|' read -r name secs cmd; do
    [ -z "$name" ] && continue
    echo "== $name (limit ${secs}s): $cmd"
    t0=$(date +%s)
    timeout -k 10 "$secs" bash -c "$cmd" > "$out/$name.txt" 2> "$out/$name.err"
    rc=$?
    echo "== $name rc=$rc in $(( $(date +%s) - t0 ))s"; tail -n 3 "$out/$name.txt"
    echo "$name rc=$rc" >> "$out/steps.log"
    if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then echo "== stopping: $name was killed or died on a signal"; tail -n 5 "$out/$name.err"; exit 1; fi
done
exit 0
