#!/bin/bash
# Round 5, third GPU visit: the whole -m gpu suite on the cleaned-up tree (halo = default 3x3 route, ABI 11), the per-layer breakdown of
# a guided forward and of a decode, the interleaved hipBLASLt yardstick, one default bench line.
cd "$(dirname "$0")/.."
TAG=${1:-r5c3}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest_all.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/log.txt
tail -4 $OUT/pytest_all.log
timeout 300 python scripts/forward_breakdown.py --top 70 > $OUT/forward_breakdown_unet.txt 2>&1; echo "breakdown unet rc=$?" | tee -a $OUT/log.txt
timeout 300 python scripts/forward_breakdown.py --decoder --top 40 > $OUT/forward_breakdown_decoder.txt 2>&1; echo "breakdown dec rc=$?" | tee -a $OUT/log.txt
timeout 400 python scripts/ws_bench.py > $OUT/ws_bench.txt 2>&1; echo "ws_bench rc=$?" | tee -a $OUT/log.txt
timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_quick.json 2> $OUT/bench_quick.err; echo "bench rc=$?" | tee -a $OUT/log.txt
head -c 600 $OUT/bench_quick.json; echo
head -12 $OUT/forward_breakdown_unet.txt
