#!/bin/bash
# Round 6, first GPU visit: the round-5 tree on this round's lease (baseline), and the fp16 floor (VERDICT r5 #5b).
cd "$(dirname "$0")/.."
TAG=${1:-r6c1}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
REPO=$(pwd)
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest_all.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/log.txt
tail -3 $OUT/pytest_all.log
cp gpurun_out/ddim50_parity_bf16.txt $OUT/ 2>/dev/null
timeout 900 python scripts/fp16_floor.py > $OUT/fp16_floor.txt 2> $OUT/fp16_floor.err; echo "fp16 floor rc=$?" | tee -a $OUT/log.txt
head -5 $OUT/fp16_floor.txt; tail -4 $OUT/fp16_floor.txt
timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/log.txt
head -c 400 $OUT/bench.json; echo
timeout 600 python scripts/forward_breakdown.py > $OUT/forward_breakdown.txt 2> $OUT/forward_breakdown.err; echo "breakdown rc=$?" | tee -a $OUT/log.txt
