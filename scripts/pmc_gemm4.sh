#!/bin/bash
# usage: scripts/pmc_gemm4.sh TAG -> gpurun_out/TAG/pmc_gemm4_{a,b,c}.json: SQ / LDS / vector-memory counters of gemm8 / gemm4 / hipBLASLt at 8192^3
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT=$ROOT/gpurun_out/${1:-pmc_gemm4}; mkdir -p $OUT; export TMPDIR=/tmp
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"
B="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM"
C="TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"
cd /tmp
for P in a b c; do
  CTRS=$A; [ $P = b ] && CTRS=$B; [ $P = c ] && CTRS=$C
  rm -rf /tmp/pmc_gemm4_$P
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $CTRS -d /tmp/pmc_gemm4_$P -o pmc -- python $ROOT/scripts/pmc_gemm4.py > $OUT/pmc_gemm4_$P.log 2>&1
  python $ROOT/scripts/pmc_dump.py "$(find /tmp/pmc_gemm4_$P -name '*.db' | head -1)" > $OUT/pmc_gemm4_$P.json 2>> $OUT/pmc_gemm4_$P.log
done
python - <<PY
import json
out = {}
for p in "abc":
    try:
        d = json.load(open("$OUT/pmc_gemm4_%s.json" % p))
    except Exception as e:
        print("pass", p, "failed:", e); continue
    for k, v in d.items():
        if not any(s in k for s in ("gemm8_kernel", "gemm4_kernel", "Cijk")):
            continue
        out.setdefault(k, {}).update(v)
for k, v in out.items():
    print(k)
    for c in sorted(v):
        print("  %-32s %d" % (c, v[c]))
PY
