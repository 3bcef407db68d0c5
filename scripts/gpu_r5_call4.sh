#!/bin/bash
# Round 5, fourth GPU visit: what the driver runs (pytest -m gpu, smoke, default bench) on the tree as it stands, the rocprofv3
# per-kernel table of a 10-step clip, the FETCH / WRITE counter passes of a guided forward.
cd "$(dirname "$0")/.."
TAG=${1:-r5c4}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
REPO=$(pwd)
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest_all.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/log.txt
tail -3 $OUT/pytest_all.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/log.txt
tail -2 $OUT/smoke.log
timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/log.txt
head -c 400 $OUT/bench.json; echo
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/profclip -o prof -- python $REPO/bench.py --steps 1 --warmup 0 --ddim-steps 10 --no-cpu-baseline --no-roofline --no-extras > $REPO/$OUT/prof.log 2>&1; python $REPO/scripts/prof_summary.py "$(find /tmp/profclip -name '*.db' | head -1)" 50 > $REPO/$OUT/prof_stats.txt 2>> $REPO/$OUT/prof.log); echo "prof rc=$?" | tee -a $OUT/log.txt
bash scripts/pmc_traffic.sh $TAG 2 > $OUT/pmc_traffic.log 2>&1; echo "pmc rc=$?" | tee -a $OUT/log.txt
tail -3 $OUT/pmc_traffic.log
