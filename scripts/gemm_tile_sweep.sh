#!/bin/bash
# microbench every shape under each forced tile family
cd "$(dirname "$0")/.."
for t in small big wide; do
  echo "=== TC_GEMM_TILE=$t"
  TC_GEMM_TILE=$t timeout 300 python scripts/gemm_bench.py 2>&1 | grep -E "linear|conv"
done
