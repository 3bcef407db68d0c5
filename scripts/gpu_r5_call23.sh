#!/bin/bash
# Round 5, GPU visit 23: the prefetch rule sweep, then the driver's three commands on the ABI-12 tree.
cd "$(dirname "$0")/.."
TAG=${1:-r5c23}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python scripts/prefetch_ab.py sweep > $OUT/prefetch_ab_sweep.txt 2> $OUT/prefetch_ab_sweep.err; echo "prefetch_ab sweep rc=$?" | tee -a $OUT/log.txt
cat $OUT/prefetch_ab_sweep.txt
t0=$(date +%s)
timeout 600 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$? $(( $(date +%s) - t0 )) s" | tee -a $OUT/log.txt
tail -4 $OUT/pytest_gpu.log
t0=$(date +%s)
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$? $(( $(date +%s) - t0 )) s" | tee -a $OUT/log.txt
tail -2 $OUT/smoke.log
t0=$(date +%s)
timeout 420 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? $(( $(date +%s) - t0 )) s" | tee -a $OUT/log.txt
cat $OUT/bench.json | cut -c1-600; tail -3 $OUT/bench.err
