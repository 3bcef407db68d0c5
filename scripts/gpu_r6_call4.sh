#!/bin/bash
# Round 6, call 4: 768-thread one-pass GroupNorm (tests, microbench, forward A/B) and the level-0 chain vs tb_fused on a second lease.
cd "$(dirname "$0")/.."
TAG=${1:-r6c4}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -p no:cacheprovider -k "groupnorm" -s > $OUT/pytest_gn.log 2>&1; echo "pytest gn rc=$?" | tee -a $OUT/log.txt
tail -3 $OUT/pytest_gn.log
timeout 300 python scripts/gn_big_bench.py > $OUT/gn_big_bench.txt 2> $OUT/gn_big_bench.err; echo "gn bench rc=$?" | tee -a $OUT/log.txt
cat $OUT/gn_big_bench.txt
timeout 600 python scripts/forward_env_ab.py TC_GN_ONEPASS_BIG 0 1 > $OUT/forward_ab_gn_big.txt 2> $OUT/forward_ab_gn.err; echo "forward ab gn rc=$?" | tee -a $OUT/log.txt
tail -7 $OUT/forward_ab_gn_big.txt
timeout 600 python scripts/forward_env_ab.py TC_TB_FUSED 1 0 > $OUT/forward_ab_tb_fused.txt 2> $OUT/forward_ab_tb.err; echo "forward ab tb rc=$?" | tee -a $OUT/log.txt
tail -7 $OUT/forward_ab_tb_fused.txt
