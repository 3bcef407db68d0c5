#!/bin/bash
# Round 6, call 5: first GPU run of the cross-attention with its query projection inside (tc_attn_d64_qproj) + forward A/B; L0 default = chain.
cd "$(dirname "$0")/.."
TAG=${1:-r6c5}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_attn_qproj.py tests/test_gpu_tb_fused.py -x -q -p no:cacheprovider -s > $OUT/pytest_qproj.log 2>&1; echo "pytest qproj rc=$?" | tee -a $OUT/log.txt
grep -E "one launch vs gemm|passed|failed|Error" $OUT/pytest_qproj.log | head -20
timeout 600 python scripts/forward_env_ab.py TC_ATTN_QPROJ 0 1 > $OUT/forward_ab_qproj.txt 2> $OUT/forward_ab_qproj.err; echo "forward ab qproj rc=$?" | tee -a $OUT/log.txt
tail -7 $OUT/forward_ab_qproj.txt
timeout 600 python scripts/forward_breakdown.py > $OUT/forward_breakdown.txt 2> $OUT/forward_breakdown.err; echo "breakdown rc=$?" | tee -a $OUT/log.txt
grep -E "attention|by operator" $OUT/forward_breakdown.txt | cut -c1-160
