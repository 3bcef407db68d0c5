#!/bin/bash
# Round 6, call 2: first GPU run of csrc/qkv_attn.hip (ABI 13) -- its tests, the microbench, the forward-level A/B, full-size parity.
cd "$(dirname "$0")/.."
TAG=${1:-r6c2}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_qkv_attn.py -x -q -p no:cacheprovider -s > $OUT/pytest_qkv_attn.log 2>&1; echo "pytest qkv_attn rc=$?" | tee -a $OUT/log.txt
tail -5 $OUT/pytest_qkv_attn.log
timeout 300 python scripts/qkv_attn_bench.py > $OUT/qkv_attn_bench.txt 2> $OUT/qkv_attn_bench.err; echo "bench rc=$?" | tee -a $OUT/log.txt
cat $OUT/qkv_attn_bench.txt
timeout 600 python scripts/forward_env_ab.py TC_QKV_ATTN 0 1 > $OUT/forward_ab_qkv_attn.txt 2> $OUT/forward_ab.err; echo "forward ab rc=$?" | tee -a $OUT/log.txt
cat $OUT/forward_ab_qkv_attn.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_models.py tests/test_gpu_torch_ops.py -x -q -p no:cacheprovider > $OUT/pytest_models.log 2>&1; echo "pytest models rc=$?" | tee -a $OUT/log.txt
tail -3 $OUT/pytest_models.log
