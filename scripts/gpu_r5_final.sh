#!/bin/bash
# Round 5, final GPU visit: the driver's commands on the final tree, the per-kernel table, and the configs[3] / configs[4] bench lines.
cd "$(dirname "$0")/.."
TAG=${1:-r5final}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
REPO=$(pwd)
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest_all.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/log.txt
tail -3 $OUT/pytest_all.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/log.txt
tail -2 $OUT/smoke.log
timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/log.txt
head -c 300 $OUT/bench.json; echo
timeout 600 python bench.py --fp8 --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench_fp8.json 2> $OUT/bench_fp8.err; echo "bench fp8 rc=$?" | tee -a $OUT/log.txt
head -c 200 $OUT/bench_fp8.json; echo
timeout 600 python bench.py --batched-decode 2 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench_b2.json 2> $OUT/bench_b2.err; echo "bench b2 rc=$?" | tee -a $OUT/log.txt
head -c 200 $OUT/bench_b2.json; echo
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/profclip -o prof -- python $REPO/bench.py --steps 1 --warmup 0 --ddim-steps 10 --no-cpu-baseline --no-roofline --no-extras > $REPO/$OUT/prof.log 2>&1; python $REPO/scripts/prof_summary.py "$(find /tmp/profclip -name '*.db' | head -1)" 50 > $REPO/$OUT/prof_stats.txt 2>> $REPO/$OUT/prof.log); echo "prof rc=$?" | tee -a $OUT/log.txt
