#!/bin/bash
# One gpurun call's worth of the round's evidence (run from the repo root on the GPU box):
#   bash tools/measure_round.sh <outdir> [stage ...]      stages: tests bench prof pmc cfgs fit opts bound shapes half stores
#                                                          round 5: sparsity lds masklayout gensk fuzzgen rehearsal refs
#                                                          round 6: retry (the launcher's second attempt, rehearsed over gloo) cbl (trunks that end in a concat)
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
    sparsity) # exact zeros of the saved tensors per element / row / block, random and trained states (trains 2 x 600 steps)
           timeout -k 10 500 python tools/saved_block_sparsity.py --state-dir $OUT/sparsity --out $OUT/saved_block_sparsity.json > $OUT/sparsity.log 2>&1 || fault sparsity
           python -c "import json; r=json.loads(open('$OUT/saved_block_sparsity.json').read()); print('sparsity', r['verdict'])" ;;
    lds)   # LDS bank conflicts of the weight-gradient kernel per ablation build; needs build.py --variant=ldsmask|ldsrows|ldsboth -DKNERF_WGRAD_ABLATE_LDS=1|2|3
           bash tools/lds_pmc.sh $OUT/lds_pmc > $OUT/wgrad_lds_conflicts.json 2> $OUT/lds_pmc.err || fault lds; rm -rf $OUT/lds_pmc/*/
           grep -E "conflict_share|^ \"" $OUT/wgrad_lds_conflicts.json ;;
    masklayout) # the conflict-free mask block (build.py --variant=maskadj -DKNERF_MASK_LAYOUT=1) against the product, ABBA order
           for v in default maskadj maskadj default default maskadj maskadj default; do
             lib=keras_nerf_amd/libknerf_hip.so; [ $v != default ] && lib=keras_nerf_amd/libknerf_hip_$v.so
             timeout -k 10 120 python tools/kbench.py --iters 20 --skip-dead-tiles 1 --lib $lib --tag $v >> $OUT/mask_layout_ab.jsonl 2>> $OUT/mask_layout_ab.err || fault "masklayout $v"
           done; grep -c kernels $OUT/mask_layout_ab.jsonl ;;
    gensk) # the general-shape kernels with and without exact dead-tile skipping on the compact scene
           for sk in "--skip-dead" ""; do KNERF_FORCE_GENERIC=1 timeout -k 10 400 python tools/convergence128.py --backend hip --scene compact $sk --lr 5e-4 --scale 1.6 --steps 600 --eval-every 200 --out $OUT/conv_generic_skip_${sk:+on}.json > $OUT/gensk_${sk:+on}.log 2>&1 || fault gensk; tail -n 1 $OUT/gensk_${sk:+on}.log | cut -c1-200; done ;;
    fuzzgen) timeout -k 10 800 python tools/fuzz_generic.py --cases 60 > $OUT/fuzz_generic.jsonl 2> $OUT/fuzz_generic.err; rc=$?; tail -n 1 $OUT/fuzz_generic.jsonl; [ $rc -eq 0 ] || fault fuzzgen ;;
    rehearsal) # an N = 3 line over gloo on one GPU (control flow and the line's diagnostics fields, not a measurement)
           KNERF_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 3 --config cfg4 --steps 10 --warmup 2 --no-cpu-baseline 2> $OUT/rehearsal.err | grep "^{" > $OUT/bench_cfg4_3ranks_gloo.json || fault rehearsal
           python -c "import json; l=json.load(open('$OUT/bench_cfg4_3ranks_gloo.json')); print('rehearsal', l['n_gpus'], l['replica_drift'], l['allreduce_us_standalone']['median'])" ;;
    retry) # the launcher's one-shot second attempt (keras_nerf_amd/parallel.py _launch_attempts): both ranks of the first set fail in the first all-reduce
           KNERF_DIST_BACKEND=gloo KNERF_BENCH_INJECT_FAILURE="*:first_all_reduce@1" timeout -k 10 300 python bench.py --gpus 2 --config cfg4 --steps 5 --warmup 2 --no-cpu-baseline 2> $OUT/retry.err | grep "^{" > $OUT/bench_cfg4_2ranks_launch_retry.json || fault retry
           python -c "import json; l=json.load(open('$OUT/bench_cfg4_2ranks_launch_retry.json')); print('retry', l['n_gpus'], l['launch_attempts'], l['ipc_mode_legacy_env'], l['replica_drift'])" ;;
    cbl)   # round 6: a concat behind the last trunk layer on the fused kernels against the general-shape kernels; needs libknerf_hip_xshape.so
           : > $OUT/cbl_kbench.jsonl
           for sh in 9,256,4,10,4 5,128,2,10,4 5,64,4,6,2; do
             timeout -k 10 120 python tools/kbench.py --lib keras_nerf_amd/libknerf_hip_xshape.so --shape $sh --tag fused_$sh >> $OUT/cbl_kbench.jsonl 2>> $OUT/cbl_kbench.err || fault "cbl $sh"
             KNERF_FORCE_GENERIC=1 timeout -k 10 120 python tools/kbench.py --lib keras_nerf_amd/libknerf_hip_xshape.so --shape $sh --tag generic_$sh >> $OUT/cbl_kbench.jsonl 2>> $OUT/cbl_kbench.err || fault "cbl generic $sh"
           done; grep -c kernels $OUT/cbl_kbench.jsonl ;;
    refs)  # the command lines the reference's source comments quote a time for, as bench configs
           for c in ref1 ref2 ref3; do timeout -k 10 200 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_$c.json 2> $OUT/bench_$c.err || fault $c; python -c "import json; l=json.load(open('$OUT/bench_$c.json')); print('$c', l['ms_per_step'], l['value'], l['vs_baseline'], l['baseline'])"; done ;;
  esac
done
exit 0
