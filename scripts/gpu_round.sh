#!/bin/bash
# One GPU-box visit: parity tests, smoke, a short bench, optional rocprof.  Everything lands in gpurun_out/.
# usage: scripts/gpu_round.sh [stage ...]   stages: ops models smoke bench benchfull prof pmc
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
STAGES="${*:-ops models smoke bench}"
echo "stages: $STAGES" | tee $OUT/round.log
rocm-smi --showproductname 2>/dev/null | head -8 >> $OUT/round.log
python -c "import os,torch;print('cpus',os.cpu_count(),'threads',torch.get_num_threads(),'gpus',torch.cuda.device_count())" >> $OUT/round.log 2>&1
for s in $STAGES; do
  t0=$(date +%s)
  case $s in
    ops)    timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -s -p no:cacheprovider > $OUT/ops.log 2>&1 ;;
    props)  timeout 600 python -m pytest tests/test_gpu_properties.py -m gpu -q -s -p no:cacheprovider > $OUT/props.log 2>&1 ;;
    models) timeout 1500 python -m pytest tests/test_gpu_models.py -m gpu -q -s -p no:cacheprovider > $OUT/models.log 2>&1 ;;
    smoke)  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 ;;
    bench)  timeout 900 python bench.py --steps 1 --warmup 1 --ddim-steps 4 --no-cpu-baseline > $OUT/bench_short.log 2>&1 ;;
    benchfull) timeout 1500 python bench.py --steps 2 --warmup 1 > $OUT/bench_full.log 2>&1 ;;
    gemmbench) timeout 600 python scripts/gemm_bench.py > $OUT/gemm_bench.log 2>&1 ;;
    benchab) (for g in "TC_NOOP=1" "${AB_ENV:-TC_NOOP=2}" "TC_NOOP=1" "${AB_ENV:-TC_NOOP=2}"; do echo "== $g"; env $g timeout 600 python bench.py --steps 1 --warmup 1 --ddim-steps 6 --no-cpu-baseline; done) > $OUT/bench_ab.log 2>&1 ;;
    tilesweep) timeout 900 bash scripts/gemm_tile_sweep.sh > $OUT/tile_sweep.log 2>&1 ;;
    pmcunet) (cd /tmp && for ctr in FETCH_SIZE WRITE_SIZE; do timeout -k 5 150 rocprofv3 --kernel-trace --pmc $ctr -d $OLDPWD/$OUT/pmcunet_$ctr -o pmc -- python $OLDPWD/scripts/pmc_unet.py 2 > $OLDPWD/$OUT/pmcunet_$ctr.log 2>&1; python $OLDPWD/scripts/pmc_parse.py "$(find $OLDPWD/$OUT/pmcunet_$ctr -name '*.db' | head -1)" 2 > $OLDPWD/$OUT/pmcunet_$ctr.json 2>> $OLDPWD/$OUT/pmcunet_$ctr.log; rm -rf $OLDPWD/$OUT/pmcunet_$ctr; done) ;;
    normbench) (for g in ${GN_SWEEP:-2048}; do TC_GN_BLOCKS=$g timeout 120 python scripts/norm_bench.py 2>&1 | grep -v amdgpu.ids; done) > $OUT/norm_bench.log 2>&1 ;;
    ordersweep) timeout 900 bash scripts/order_sweep.sh > $OUT/order_sweep.log 2>&1 ;;
    prof)   (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof -o prof -- python $OLDPWD/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OLDPWD/$OUT/prof.log 2>&1; python $OLDPWD/scripts/prof_summary.py "$(find $OLDPWD/$OUT/prof -name '*.db' | head -1)" > $OLDPWD/$OUT/prof_stats.txt 2>> $OLDPWD/$OUT/prof.log; rm -rf $OLDPWD/$OUT/prof) ;;
    *) echo "unknown stage $s" ;;
  esac
  rc=$?
  echo "stage $s rc=$rc $(( $(date +%s) - t0 ))s" | tee -a $OUT/round.log
done
tail -3 $OUT/*.log 2>/dev/null | tail -60
