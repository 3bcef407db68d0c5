#!/bin/bash
# Round 5, GPU visit 25: is the A operand a launch has just written warm for the next launch? (scripts/fresh_activation_probe.py)
cd "$(dirname "$0")/.."
TAG=${1:-r5c25}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python scripts/fresh_activation_probe.py > $OUT/fresh_activation_probe.txt 2> $OUT/fresh_activation_probe.err; echo "fresh_activation_probe rc=$?" | tee -a $OUT/log.txt
cat $OUT/fresh_activation_probe.txt; tail -4 $OUT/fresh_activation_probe.err
