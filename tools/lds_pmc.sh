#!/bin/bash
# LDS bank conflicts of the weight-gradient kernel, per ablation build (tests/test_lds_bank_model.py
# holds the analytic attribution).  One rocprofv3 --pmc pass per library (counters only: never combined with tracing domains), the
# program directly behind `--`.
#   bash tools/lds_pmc.sh <outdir> [variant ...]     variants: default ldsmask ldsrows ldsboth (build.py --variant=<v> -DKNERF_WGRAD_ABLATE_LDS=1|2|3)
set -u
OUT=${1:-gpurun_out/lds_pmc}; shift || true
VARIANTS=${*:-default ldsmask ldsrows ldsboth}
export TMPDIR=/tmp
mkdir -p $OUT
for v in $VARIANTS; do
  lib=keras_nerf_amd/libknerf_hip.so; [ $v != default ] && lib=keras_nerf_amd/libknerf_hip_$v.so
  rm -rf $OUT/$v
  timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/$v -- \
      python3 tools/kbench.py --iters 2 --skip-dead-tiles 1 --lib $lib --tag $v > $OUT/$v.log 2>&1 || { echo "pass $v failed"; tail -5 $OUT/$v.log; exit 1; }
done
python3 - "$OUT" $VARIANTS <<'PY'
import csv, glob, json, sys, collections
out, variants = sys.argv[1], sys.argv[2:]
rep = {}
for v in variants:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/{v}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("knerf::", "")
            if "wgrad_kernel" in k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    rep[v] = {}
    for k, cs in sorted(agg.items()):
        # the coarse and the fine launch share a kernel name per net; report the mean per dispatch
        row = {c: round(sum(x) / len(x), 1) for c, x in sorted(cs.items())}
        if row.get("SQ_LDS_IDX_ACTIVE"):
            row["conflict_share"] = round(row.get("SQ_LDS_BANK_CONFLICT", 0.0) / row["SQ_LDS_IDX_ACTIVE"], 4)
        rep[v][k] = row
print(json.dumps(rep, indent=1))
PY
