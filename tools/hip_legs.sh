#!/bin/bash
# One gpurun call of extra HIP convergence legs, one after the other (each ~75 s):  bash tools/hip_legs.sh <first> <last> [budget-s]
# -> gpurun_out/conv128_hip<k>.json.  Stops starting new legs once the budget is used.
set -u
A=$1; B=$2; BUD=${3:-1000}
t0=$(date +%s)
for k in $(seq $A $B); do
  [ $(( $(date +%s) - t0 )) -gt $BUD ] && break
  timeout -k 10 300 python tools/convergence128.py --backend hip --steps 2000 --eval-every 100 --lr 5e-4 --scale 1.6 \
      --out gpurun_out/conv128_hip$k.json > gpurun_out/conv128_hip$k.log 2>&1 || { echo "leg $k failed"; tail -n 5 gpurun_out/conv128_hip$k.log; exit 1; }
  tail -n 1 gpurun_out/conv128_hip$k.log | cut -c1-160
done
exit 0
