#!/bin/bash
# Round 5, second GPU visit: the cooperative GroupNorm (never run), the 3x3-only halo routing at clip level, hipBLASLt kernel names.
cd "$(dirname "$0")/.."
TAG=${1:-r5c2}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
REPO=$(pwd)
timeout 300 python -m pytest tests/test_gpu_gn_coop.py -m gpu -x -q -p no:cacheprovider > $OUT/pytest_gn_coop.log 2>&1; echo "gn_coop tests rc=$?" | tee -a $OUT/log.txt
tail -5 $OUT/pytest_gn_coop.log
if tail -3 $OUT/pytest_gn_coop.log | grep -q "passed" && ! tail -3 $OUT/pytest_gn_coop.log | grep -q "failed\|error"; then COOP_OK=1; else COOP_OK=0; fi
timeout 300 python scripts/norm_bench.py > $OUT/norm_bench.txt 2>&1; echo "norm_bench rc=$?" | tee -a $OUT/log.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/hbl -o hbl -- python $REPO/scripts/hipblaslt_names.py > $REPO/$OUT/hipblaslt_names.txt 2>&1; python $REPO/scripts/prof_summary.py "$(find /tmp/hbl -name '*.db' | head -1)" 30 > $REPO/$OUT/hipblaslt_kernels.txt 2>&1)
echo "hipblaslt rc=$?" | tee -a $OUT/log.txt
# clip-level A/B, alternating: (coop, halo3x3, tall)
run() { echo "== TC_GN_COOP=$1 TC_CONV_HALO=$2 TC_CONV_HALO_TALL=$3"; TC_GN_COOP=$1 TC_CONV_HALO=$2 TC_CONV_HALO_T3=0 TC_CONV_HALO_TALL=$3 timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['stage_ms_per_clip'], 'gemm frac', d['roofline']['frac'], 'gn ms', d['roofline_hbm']['ms_per_unet_fwd_b2'], 'gn frac', d['roofline_hbm']['frac'])"; }
if [ $COOP_OK = 1 ]; then ARMS="0,0,0 1,0,0 0,1,0 1,1,0 0,0,0 1,0,0 1,1,1 1,1,0"; else ARMS="0,0,0 0,1,0 0,0,0 0,1,0 0,1,1"; fi
(for a in $ARMS; do IFS=, read x y z <<< "$a"; run $x $y $z; done) > $OUT/clip_ab.txt 2>&1
cat $OUT/clip_ab.txt
