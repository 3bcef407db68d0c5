#!/bin/bash
# Round-5 GPU visits, as PLANNED at the end of round 4 (the visits that were actually made: scripts/gpu_r5_call1.sh .. call4.sh;
# the `unver` gate is gone: those tests are part of the default -m gpu run now).  usage: scripts/gpu_round5.sh TAG stage...     (everything under its own `timeout`)
#   unver   : the GATED tests of kernels written without a GPU (conv_halo.hip) -- run FIRST, alone, short timeout
#   halo    : kernel-level A/B of TC_CONV_HALO on the UNet's convolution shapes
#   haloclip: clip-level A/B (alternating) of TC_CONV_HALO=1 against the default routing
#   fuseunet: the full-size / tiny model parity tests with TC_GN_FUSE=1 TC_CONV_HALO=1 (GroupNorm inside the convolutions)
#   fuseclip: clip-level A/B of (halo, fuse) = (0,0) (1,1) (0,0) (1,1) (1,0)
#   pmchalo : SQ / LDS / L2 counters of the level-0 3x3 convolution on gemm16 vs the halo kernel (three --pmc passes)
#   gn      : GroupNorm operator timings on the UNet's shapes (the SiLU change of round 4's last session was never timed)
#   all / smoke / bench / benchq / prof : as in scripts/gpu_round4.sh
set -u
cd "$(dirname "$0")/.."
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
REPO=$(pwd)
for s in "$@"; do
  t0=$(date +%s)
  case $s in
    unver)   timeout 300 python -m pytest tests/test_gpu_conv_halo.py -m gpu -x -q -p no:cacheprovider > $OUT/pytest_unverified.log 2>&1 ;;
    halo)    timeout 300 python scripts/conv_halo_bench.py > $OUT/conv_halo_bench.txt 2>&1 ;;
    haloclip) (for v in 0 1 0 1; do echo "== TC_CONV_HALO=$v"; TC_CONV_HALO=$v timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['stage_ms_per_clip'])"; done) > $OUT/conv_halo_clip_ab.txt 2>&1 ;;
    gn)      timeout 300 python scripts/norm_bench.py > $OUT/norm_bench.txt 2>&1 ;;
    fuseunet) TC_GN_FUSE=1 TC_CONV_HALO=1 timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_models.py -m gpu -x -q -p no:cacheprovider > $OUT/pytest_fused_unet.log 2>&1 ;;
    fuseclip) (for v in "0 0" "1 1" "0 0" "1 1" "1 0"; do set -- $v; echo "== TC_CONV_HALO=$1 TC_GN_FUSE=$2"; TC_CONV_HALO=$1 TC_GN_FUSE=$2 timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['stage_ms_per_clip'], 'gemm frac', d['roofline']['frac'], 'gn ms', d['roofline_hbm']['ms_per_unet_fwd_b2'])"; done) > $OUT/fuse_clip_ab.txt 2>&1 ;;
    all)     timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest_all.log 2>&1 ;;
    smoke)   timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 ;;
    bench)   timeout 900 python bench.py --steps 10 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err ;;
    benchq)  timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_quick.json 2> $OUT/bench_quick.err ;;
    prof)    (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/profclip -o prof -- python $REPO/bench.py --steps 1 --warmup 0 --ddim-steps 10 --no-cpu-baseline --no-roofline > $REPO/$OUT/prof.log 2>&1; python $REPO/scripts/prof_summary.py "$(find /tmp/profclip -name '*.db' | head -1)" 50 > $REPO/$OUT/prof_stats.txt 2>> $REPO/$OUT/prof.log) ;;
    pmchalo) bash scripts/pmc_halo.sh > $OUT/pmc_halo.log 2>&1 ;;
    halodbg) (for a in "3x3" "t3" "3x3 2 20 32 128 160 tall" "3x3 2 10 16 256 160 ksplit" "3x3 2 20 32 128 160 gn"; do timeout 120 python scripts/conv_halo_debug.py $a; done) > $OUT/conv_halo_debug.txt 2>&1 ;;
    py:*)    timeout 600 python ${s#py:} > $OUT/$(basename ${s#py:} .py).txt 2>&1 ;;
    *) echo "unknown stage $s" ;;
  esac
  echo "stage $s rc=$? $(( $(date +%s) - t0 ))s" | tee -a $OUT/round5.log
done
tail -3 $OUT/*.log $OUT/*.txt 2>/dev/null | tail -60
