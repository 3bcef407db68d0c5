#!/bin/bash
# Same-process forward A/B of existing switches on the final tree (no new code): is each default still the faster arm?
cd "$(dirname "$0")/.."
TAG=${1:-r6sweep}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for spec in "TC_FF_FUSED 1 0" "TC_PREFETCH 0 1" "TC_CONV_HALO 0 1" "TC_GEMM8 0 1" "TC_FUSE_LN 0 1"; do
  set -- $spec
  timeout 400 python scripts/forward_env_ab.py $1 $2 $3 --rounds 3 > $OUT/ab_$1.txt 2> $OUT/ab_$1.err
  echo "$spec: $(tail -1 $OUT/ab_$1.txt)" | tee -a $OUT/summary.txt
done
