#!/bin/bash
# Round 5, fifth GPU visit: the driver's commands on the tree with the custom-op binding as the default.
cd "$(dirname "$0")/.."
TAG=${1:-r5c5}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest_all.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/log.txt
tail -3 $OUT/pytest_all.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/log.txt
tail -2 $OUT/smoke.log
timeout 1200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/log.txt
head -c 300 $OUT/bench.json; echo; tail -3 $OUT/bench.err
