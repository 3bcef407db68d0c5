#!/bin/bash
# Round 5, GPU visit 22: ABI 12 (weight prefetch on the norm launches): its tests, then the forward-level A/B.
cd "$(dirname "$0")/.."
TAG=${1:-r5c22}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_guard.py tests/test_gpu_torch_ops.py -m gpu -x -q -p no:cacheprovider -k "prefetch or groupnorm or layernorm or bit_identical or torch" > $OUT/pytest_prefetch.log 2>&1; echo "prefetch tests rc=$?" | tee -a $OUT/log.txt
tail -6 $OUT/pytest_prefetch.log
timeout 300 python scripts/prefetch_ab.py > $OUT/prefetch_ab.txt 2> $OUT/prefetch_ab.err; echo "prefetch_ab rc=$?" | tee -a $OUT/log.txt
cat $OUT/prefetch_ab.txt; tail -4 $OUT/prefetch_ab.err
