#!/bin/bash
# Round 5, GPU visit 21: does a read of the weights one launch ahead remove the cold-weight penalty? (scripts/prefetch_premise_probe.py)
cd "$(dirname "$0")/.."
TAG=${1:-r5c21}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 420 python scripts/prefetch_premise_probe.py > $OUT/prefetch_premise_probe.txt 2> $OUT/prefetch_premise_probe.err; echo "prefetch_premise_probe rc=$?" | tee -a $OUT/log.txt
cat $OUT/prefetch_premise_probe.txt; tail -5 $OUT/prefetch_premise_probe.err
