#!/bin/bash
# The two shipped round-6 changes, A/B'd again on another lease: (i) tc_temporal_qkv_attn at levels 1-3 with level 0 on the round-4
# single launch in BOTH arms (the +1.66 % measurement's configuration), (ii) level 0: single launch (A) vs the chain (B).
cd "$(dirname "$0")/.."
TAG=${1:-r6ab}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
TC_TB_FUSED=1 timeout 600 python scripts/forward_env_ab.py TC_QKV_ATTN 0 1 > $OUT/forward_ab_qkv_attn.txt 2> $OUT/ab1.err; echo "ab qkv rc=$?" | tee -a $OUT/log.txt
tail -7 $OUT/forward_ab_qkv_attn.txt
timeout 600 python scripts/forward_env_ab.py TC_TB_FUSED 1 0 > $OUT/forward_ab_l0_chain.txt 2> $OUT/ab2.err; echo "ab l0 rc=$?" | tee -a $OUT/log.txt
tail -7 $OUT/forward_ab_l0_chain.txt
