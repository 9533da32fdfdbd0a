#!/bin/bash
# One gpurun call's worth of the round's evidence (run from the repo root on the GPU box):
#   bash tools/measure_round.sh <outdir> [stage ...]      stages: tests bench prof pmc cfgs fit opts bound shapes half stores
# Every stage writes small files under <outdir>; profiles/ holds the copies that are committed (profiles/README.md).
set -u
OUT=${1:-gpurun_out/measure}; shift || true
STAGES=${*:-tests bench prof pmc cfgs fit opts}
mkdir -p $OUT
export TMPDIR=/tmp
fault() { echo "stage $1 failed or faulted: stopping"; rm -f gpucore.* core*; exit 1; }
for st in $STAGES; do
  case $st in
    tests) timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; rc=$?; tail -n 3 $OUT/pytest_gpu.log; [ $rc -eq 0 ] || fault tests ;;
    bench) timeout -k 10 300 python bench.py > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err || fault bench; python -c "import json; l=json.load(open('$OUT/bench_cfg2.json')); print('cfg2', l['ms_per_step'], l['value'], l['roofline']['frac'], l['metrics_ms_per_step'], l['dead_tile_frac'])" ;;
    prof)  rm -rf $OUT/prof; (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof -- python3 $OLDPWD/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OLDPWD/$OUT/bench_under_rocprofv3.json 2> $OLDPWD/$OUT/prof.err) || fault prof
           f=$(find $OUT/prof -name "*kernel_stats.csv" | head -n 1); cp "$f" $OUT/kernel_stats.csv; head -n 12 $OUT/kernel_stats.csv | cut -c1-150; rm -rf $OUT/prof ;;
    pmc)   # the instantiations bench.py runs by default: list mode (skip_dead_tiles = 1)
           bash tools/pmc.sh $OUT/pmc --skip-dead-tiles 1 > $OUT/pmc.log 2>&1 || fault pmc; python tools/pmc_report.py $OUT/pmc act118_dz114 1 > $OUT/pmc_traffic.json; rm -rf $OUT/pmc; python -c "import json; r=json.load(open('$OUT/pmc_traffic.json')); print({k: v.get('hbm_bytes_per_launch') for k, v in r.items() if isinstance(v, dict) and 'hbm_bytes_per_launch' in v})" ;;
    cfgs)  for c in cfg3 cfg4 cfg5; do timeout -k 10 200 python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_$c.json 2> $OUT/bench_$c.err || fault $c; python -c "import json; l=json.load(open('$OUT/bench_$c.json')); print('$c', l['ms_per_step'], l['value'])"; done ;;
    fit)   for c in cfg2 cfg4; do timeout -k 10 200 python bench.py --mode fit --config $c > $OUT/fit_$c.json 2> $OUT/fit_$c.err || fault fit_$c; python -c "import json; l=json.load(open('$OUT/fit_$c.json')); print('fit $c', l['ms_per_step'], l['train_step_ms'], l['fit_vs_train_step'], l['metrics_ms_per_step'])"; done ;;
    opts)  for o in "--skip-dead-tiles 0" "--skip-dead-tiles 1" "--deterministic 1" "--skip-dead-tiles 0"; do n=$(echo $o | tr -d ' -'); timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $o > $OUT/opt_$n.json 2> $OUT/opt_$n.err || fault "$o"; python -c "import json; l=json.load(open('$OUT/opt_$n.json')); print('$o', l['ms_per_step'], l['roofline']['kernel_ms_per_step'])"; done ;;
    bound) # upper bound of a forward that does not save what the backward skips; needs libknerf_hip_nofwdstores.so
           timeout -k 10 600 python tools/fwd_save_bound.py --weights $OUT/fwd_bound_w_step600.npz > $OUT/fwd_save_bound.json 2> $OUT/fwd_save_bound.err || fault bound
           python -c "import json; r=json.loads(open('$OUT/fwd_save_bound.json').read().splitlines()[-1]); print('bound', r['summary'])" ;;
    half)  # ceiling of 8-bit saved tensors: every saved activation / dZ block written and read at HALF its bytes, same
           # instruction counts; needs `build.py --variant=halfsaved -DKNERF_ABLATE_HALF_SAVED`.  Alternating with the default library.
           # halfsaved2 (+ -DKNERF_ABLATE_HALF_WGRAD_MATH): also half the transposed reads and MFMAs per tile in the weight-gradient kernel
           for k in 1 2; do for v in default halfsaved halfsaved2; do
             lib=keras_nerf_amd/libknerf_hip.so; [ $v != default ] && lib=keras_nerf_amd/libknerf_hip_$v.so
             KNERF_LIB=$PWD/$lib timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --ignore-nonfinite --skip-dead-tiles 0 > $OUT/half_${v}_$k.json 2> $OUT/half_${v}_$k.err || fault "half $v"
             python -c "import json; l=json.load(open('$OUT/half_${v}_$k.json')); print('$v', l['ms_per_step'], l['roofline']['kernel_ms_per_step'])"
           done; done ;;
    stores) # what a chain kernel's saved-tensor stores cost and where: issued with no lane active (storeexec0: issue + vmcnt only),
           # or into 256 cache-resident tile slots (storel2: the whole on-chip path, little HBM); `build.py --variant=storeexec0
           # -DKNERF_ABLATE_STORE_EXEC0`, `--variant=storel2 -DKNERF_ABLATE_STORE_L2`.  Alternating with the default library.
           for k in 1 2; do for v in default storeexec0 storel2; do
             lib=keras_nerf_amd/libknerf_hip.so; [ $v != default ] && lib=keras_nerf_amd/libknerf_hip_$v.so
             KNERF_LIB=$PWD/$lib timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --ignore-nonfinite --skip-dead-tiles 0 > $OUT/stores_${v}_$k.json 2> $OUT/stores_${v}_$k.err || fault "stores $v"
             python -c "import json; l=json.load(open('$OUT/stores_${v}_$k.json')); print('$v', l['ms_per_step'], l['roofline']['kernel_ms_per_step'])"
           done; done ;;
    shapes) # the round-4 fused shapes (width 64, pos_emb_dir 8 / 6) against the general-shape kernels; needs libknerf_hip_xshape.so
           : > $OUT/shapes_kbench.jsonl
           for sh in 8,256,4,10,4 8,64,4,10,4 4,64,2,10,4 8,256,4,10,8 8,128,4,10,6; do
             timeout -k 10 120 python tools/kbench.py --lib keras_nerf_amd/libknerf_hip_xshape.so --shape $sh --tag fused_$sh >> $OUT/shapes_kbench.jsonl 2>> $OUT/shapes_kbench.err || fault "shapes $sh"
             KNERF_FORCE_GENERIC=1 timeout -k 10 120 python tools/kbench.py --lib keras_nerf_amd/libknerf_hip_xshape.so --shape $sh --tag generic_$sh >> $OUT/shapes_kbench.jsonl 2>> $OUT/shapes_kbench.err || fault "shapes generic $sh"
           done; python -c "
import json
for l in open('$OUT/shapes_kbench.jsonl'):
    if l.startswith('{'):
        r=json.loads(l); print(r.get('tag'), r.get('train_chunk_ms'), r.get('Mrs_per_s'))" ;;
  esac
done
exit 0
