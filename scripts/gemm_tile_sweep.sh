#!/bin/bash
# microbench every shape under each forced tile family (the env override is read once per process);
# "auto" = the library's own heuristics, "nosplit" = heuristics with split-K disabled
cd "$(dirname "$0")/.."
for t in ${SWEEP_TILES:-22 21 12 11 wide}; do
  echo "=== TC_GEMM_TILE=$t"
  case $t in
    auto)    timeout 300 python scripts/gemm_bench.py --more --no-attn 2>&1 | grep -E "linear|conv" ;;
    nosplit) TC_GEMM_SPLITK=0 timeout 300 python scripts/gemm_bench.py --more --no-attn 2>&1 | grep -E "linear|conv" ;;
    *)       TC_GEMM_TILE=$t timeout 300 python scripts/gemm_bench.py --more --no-attn 2>&1 | grep -E "linear|conv" ;;
  esac
done
