#!/bin/bash
# Runs GPU steps one after another inside one gpurun call: `bash tools/gpu_steps.sh "<cmd 1>" "<cmd 2>" ...`.
# A step that FAILS (assertion, non-zero exit) is reported and the next one still runs; a step that was KILLED at its time limit
# (exit 124 / 137: it may have hung the GPU) ends the call -- no further GPU step is started behind it.
export TMPDIR=/tmp
n=0
for cmd in "$@"; do
  n=$((n + 1))
  echo "=== step $n: $cmd"
  bash -o pipefail -c "$cmd"; rc=$?
  echo "=== step $n rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $n was killed at its limit: stopping"; rm -f gpucore.* core*; exit $rc; fi
done
exit 0
