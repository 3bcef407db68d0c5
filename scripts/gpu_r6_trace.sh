#!/bin/bash
# Per-kernel trace of the final tree: a warm-up clip (graphs captured) + one traced clip of 10 DDIM steps + 2 decodes, summarised by scripts/prof_summary.py.
cd "$(dirname "$0")/.."
TAG=${1:-r6trace}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
REPO=$(pwd)
(cd /tmp && rm -rf /tmp/profclip && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/profclip -o prof -- python $REPO/bench.py --steps 1 --warmup 1 --ddim-steps 10 --no-cpu-baseline --no-roofline --no-extras > $REPO/$OUT/prof.log 2>&1; python $REPO/scripts/prof_summary.py "$(find /tmp/profclip -name '*.db' | head -1)" 60 $REPO/$OUT/prof.log 10 > $REPO/$OUT/prof_stats.txt 2>> $REPO/$OUT/prof.log); echo "prof rc=$?" | tee -a $OUT/log.txt
head -4 $OUT/prof_stats.txt | cut -c1-250; grep -E "GEMM family|whole window|per DDIM step" $OUT/prof_stats.txt | cut -c1-300
tail -3 $OUT/prof_stats.txt | cut -c1-300
