#!/bin/bash
# One gpurun call of the extra fp32 convergence legs (tools/convergence128.py --backend fp32 --perturb K): the legs named on the command
# line share the GPU; each resumes from tools/_state/fp32_pK.pt (carried in the repo snapshot, git-ignored) and checkpoints into
# gpurun_out/st/ (merged back; copy it to tools/_state/ before the next call).   bash tools/fp32_legs.sh <budget-s> K [K ...]
set -u
B=$1; shift
mkdir -p gpurun_out/st
pids=""
for k in "$@"; do
  [ -f tools/_state/fp32_p$k.pt ] && cp tools/_state/fp32_p$k.pt gpurun_out/st/fp32_p$k.pt
  timeout -k 10 $((B + 150)) python tools/convergence128.py --backend fp32 --steps 2000 --eval-every 100 --lr 5e-4 --scale 1.6 --perturb $k \
      --out gpurun_out/conv128_fp32_p$k.json --state gpurun_out/st/fp32_p$k.pt --budget-s $B > gpurun_out/conv128_fp32_p$k.log 2>&1 &
  pids="$pids $!"
done
rc=0
for p in $pids; do wait $p || rc=1; done
tail -n 2 gpurun_out/conv128_fp32_p*.log
du -sh gpurun_out
exit $rc
