#!/bin/bash
# usage: scripts/pmc_halo.sh   -> gpurun_out/pmc_halo_{a,b,c}.json: SQ / LDS / L2 counters of the level-0 3x3 convolution on
# gemm16 and on the halo-patch kernel (counters in their own runs with --kernel-trace only, as the pool requires)
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"
B="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM"
C="TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"
cd /tmp
for P in a b c; do
  CTRS=$A; [ $P = b ] && CTRS=$B; [ $P = c ] && CTRS=$C
  rm -rf /tmp/pmc_halo_$P
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $CTRS -d /tmp/pmc_halo_$P -o pmc -- python $ROOT/scripts/pmc_halo.py > $OUT/pmc_halo_$P.log 2>&1
  for K in gemm16 conv_halo; do
    python $ROOT/scripts/pmc_dump.py "$(find /tmp/pmc_halo_$P -name '*.db' | head -1)" $K > $OUT/pmc_halo_${P}_$K.json 2>> $OUT/pmc_halo_$P.log
  done
done
