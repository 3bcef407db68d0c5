#!/bin/bash
# (measured and NOT shipped: plan 2 was -0.27 % where plan 1 was -0.10 %, on a slow-class lease -- profiles/r05_prefetch_plan2_ab_slow_lease.txt; the patch: scripts/experiments/prefetch_plan2.patch.txt)
# Round 5, GPU visit 27: prefetch plan 2 (a level-3 ResBlock's per-frame norms carry the whole block's weights) against plan 1, then the model parity tests.
cd "$(dirname "$0")/.."
TAG=${1:-r5c27}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 200 python scripts/prefetch_ab.py plans > $OUT/prefetch_ab_plans.txt 2> $OUT/prefetch_ab_plans.err; echo "prefetch_ab plans rc=$?" | tee -a $OUT/log.txt
cat $OUT/prefetch_ab_plans.txt; tail -3 $OUT/prefetch_ab_plans.err
timeout 170 python -m pytest tests/test_gpu_models.py -m gpu -x -q -p no:cacheprovider > $OUT/pytest_models.log 2>&1; echo "model tests rc=$?" | tee -a $OUT/log.txt
tail -3 $OUT/pytest_models.log
