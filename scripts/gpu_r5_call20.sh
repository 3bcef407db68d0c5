#!/bin/bash
# Round 5, GPU visit 20: cold operands / kernel alternation against the warm per-shape loop (scripts/cold_operand_probe.py).
cd "$(dirname "$0")/.."
TAG=${1:-r5c20}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 420 python scripts/cold_operand_probe.py > $OUT/cold_operand_probe.txt 2> $OUT/cold_operand_probe.err; echo "cold_operand_probe rc=$?" | tee -a $OUT/log.txt
cat $OUT/cold_operand_probe.txt; tail -5 $OUT/cold_operand_probe.err
