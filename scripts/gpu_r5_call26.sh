#!/bin/bash
# Round 5, GPU visit 26: rocprofv3 kernel trace of one clip (10 DDIM steps + both decodes) on the ABI-12 tree.
cd "$(dirname "$0")/.."
REPO=$(pwd); TAG=${1:-r5c26}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/profclip -o prof -- python $REPO/bench.py --steps 1 --warmup 0 --ddim-steps 10 --no-cpu-baseline --no-roofline --no-extras > $REPO/$OUT/prof.log 2>&1; python $REPO/scripts/prof_summary.py "$(find /tmp/profclip -name '*.db' | head -1)" 50 > $REPO/$OUT/prof_stats.txt 2>> $REPO/$OUT/prof.log); echo "prof rc=$?" | tee -a $OUT/log.txt
head -30 $OUT/prof_stats.txt | cut -c1-150
