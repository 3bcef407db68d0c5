#!/bin/bash
# The driver's round-end commands on the final tree: pytest -m gpu, smoke(), bench.py.
cd "$(dirname "$0")/.."
TAG=${1:-r6drv}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
t0=$(date +%s); timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest_all.log 2>&1; echo "pytest rc=$? $(( $(date +%s) - t0 )) s" | tee -a $OUT/log.txt
tail -3 $OUT/pytest_all.log
t0=$(date +%s); timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$? $(( $(date +%s) - t0 )) s" | tee -a $OUT/log.txt
tail -2 $OUT/smoke.log
t0=$(date +%s); timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? $(( $(date +%s) - t0 )) s" | tee -a $OUT/log.txt
head -c 300 $OUT/bench.json; echo
