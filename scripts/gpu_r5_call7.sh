#!/bin/bash
# Round 5, seventh GPU visit: the four-wave GEMM (csrc/gemm4.hip), never run: tests under a timeout, then the A/B + yardstick.
cd "$(dirname "$0")/.."
TAG=${1:-r5c7}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_gemm4.py -m gpu -x -q -p no:cacheprovider > $OUT/pytest_gemm4.log 2>&1; echo "gemm4 tests rc=$?" | tee -a $OUT/log.txt
tail -12 $OUT/pytest_gemm4.log
timeout 300 python scripts/gemm4_bench.py > $OUT/gemm4_bench.txt 2>&1; echo "gemm4_bench rc=$?" | tee -a $OUT/log.txt
cat $OUT/gemm4_bench.txt | tail -14
