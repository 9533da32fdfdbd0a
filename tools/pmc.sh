#!/bin/bash
# Collects PMC counters for the hot kernels in separate rocprofv3 passes (never combined with tracing domains).
# usage (on the GPU box, from the repo root): bash tools/pmc.sh <outdir> [kbench args]
set -e
OUT=${1:-gpurun_out/pmc}; shift || true
export TMPDIR=/tmp
mkdir -p $OUT
run() {  # name, counters...
  name=$1; shift
  rm -rf $OUT/$name
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 tools/kbench.py --iters 2 "${KARGS[@]}" > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; }
}
KARGS=("$@")
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU
run sq2 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR
run fetch FETCH_SIZE TCC_HIT_sum
run write WRITE_SIZE TCC_MISS_sum TCC_REQ_sum
run grbm GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
find $OUT -name "*counter_collection.csv" | head
