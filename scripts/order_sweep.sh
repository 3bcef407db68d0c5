#!/bin/bash
# chunk width / threshold sweep of the L2-aware tile walk on the wide-N shapes (env read once per process)
cd "$(dirname "$0")/.."
for cfg in "0 4" "4 4" "8 4" "16 4" "8 2" "8 1" "4 1"; do
  set -- $cfg
  echo "=== ORDER=$1 MIB=$2"
  TC_GEMM_ORDER=$1 TC_GEMM_ORDER_MIB=$2 timeout 120 python scripts/gemm_bench.py --more --no-attn 2>&1 | grep -E "geglu|qkv|proj|ff2|L2 res|L1 res "
done
