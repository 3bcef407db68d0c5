#!/bin/bash
# Round 6, call 3: full GPU suite + bench + per-kernel trace on the ABI 13 tree (qkv + temporal attention as one launch by default).
cd "$(dirname "$0")/.."
TAG=${1:-r6c3}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
REPO=$(pwd)
timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest_all.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/log.txt
tail -3 $OUT/pytest_all.log
cp gpurun_out/ddim50_parity_*.txt $OUT/ 2>/dev/null
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -p no:cacheprovider -s 2>&1 | grep -E "rel-L2|floor|passed|failed" > $OUT/fullsize_numbers.txt; echo "fullsize rc=$?" | tee -a $OUT/log.txt
timeout 600 python scripts/forward_env_ab.py TC_TB_FUSED 0 1 --rounds 3 > $OUT/forward_ab_tb_fused.txt 2>&1; echo "ab tb rc=$?" | tee -a $OUT/log.txt
head -3 $OUT/forward_ab_tb_fused.txt; tail -1 $OUT/forward_ab_tb_fused.txt
timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/log.txt
head -c 300 $OUT/bench.json; echo
timeout 600 python scripts/forward_breakdown.py > $OUT/forward_breakdown.txt 2> $OUT/forward_breakdown.err; echo "breakdown rc=$?" | tee -a $OUT/log.txt
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/profclip -o prof -- python $REPO/bench.py --steps 1 --warmup 0 --ddim-steps 10 --no-cpu-baseline --no-roofline --no-extras > $REPO/$OUT/prof.log 2>&1; python $REPO/scripts/prof_summary.py "$(find /tmp/profclip -name '*.db' | head -1)" 60 > $REPO/$OUT/prof_stats.txt 2>> $REPO/$OUT/prof.log); echo "prof rc=$?" | tee -a $OUT/log.txt
