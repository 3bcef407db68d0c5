#!/bin/bash
# Round-4 GPU visit.  usage: scripts/gpu_round4.sh TAG stage...
set -u
cd "$(dirname "$0")/.."
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
REPO=$(pwd)
for s in "$@"; do
  t0=$(date +%s)
  case $s in
    all)     timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest_all.log 2>&1 ;;
    addr)    timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_guard.py tests/test_gpu_gemm_ws.py tests/test_gpu_models.py tests/test_gpu_properties.py tests/test_gpu_fullsize.py -m gpu -x -q -p no:cacheprovider -k "not fp8" > $OUT/pytest_addr.log 2>&1 ;;
    smoke)   timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 ;;
    bench)   timeout 900 python bench.py --steps 10 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err ;;
    benchq)  timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_quick.json 2> $OUT/bench_quick.err ;;
    benchbd) timeout 900 python bench.py --steps 5 --warmup 1 --batched-decode 2 --no-cpu-baseline --no-roofline > $OUT/bench_batched_decode_b2.json 2> $OUT/bench_bd.err ;;
    prof)    (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/profclip -o prof -- python $REPO/bench.py --steps 1 --warmup 0 --ddim-steps 10 --no-cpu-baseline --no-roofline > $REPO/$OUT/prof.log 2>&1; python $REPO/scripts/prof_summary.py "$(find /tmp/profclip -name '*.db' | head -1)" 50 > $REPO/$OUT/prof_stats.txt 2>> $REPO/$OUT/prof.log) ;;
    opsq)    timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_guard.py -m gpu -x -q -p no:cacheprovider -k "not beyond_2gib" > $OUT/pytest_ops.log 2>&1 ;;
    shareab) (for v in 1 0 1 0; do echo "== TC_CFG_SHARE=$v"; TC_CFG_SHARE=$v timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['stage_ms_per_clip'])"; done) > $OUT/share_ab.txt 2>&1 ;;
    benchfp8) timeout 600 python bench.py --fp8 --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/bench_fp8.json 2> $OUT/bench_fp8.err ;;
    rest)    timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_properties.py tests/test_gpu_real_yaml.py tests/test_gpu_torch_ops.py tests/test_two_clips.py -m gpu -x -q -p no:cacheprovider > $OUT/pytest_rest.log 2>&1 ;;
    pmc)     bash scripts/pmc_traffic.sh $TAG 2 > $OUT/pmc_traffic.log 2>&1 ;;
    wsbench) timeout 300 python scripts/ws_bench.py > $OUT/ws_bench.txt 2>&1 ;;
    py:*)    timeout 600 python ${s#py:} > $OUT/$(basename ${s#py:} .py).txt 2>&1 ;;
    *) echo "unknown stage $s" ;;
  esac
  echo "stage $s rc=$? $(( $(date +%s) - t0 ))s" | tee -a $OUT/round4.log
done
tail -3 $OUT/*.log 2>/dev/null | tail -40
