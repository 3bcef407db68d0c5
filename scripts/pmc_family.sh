#!/bin/bash
# In-situ counters of one guided (B = 2) UNet forward by kernel family: matrix-pipe busy fraction, wait fraction, L1 -> L2 read
# requests per second.  Two rocprofv3 --kernel-trace --pmc passes (never with sys/hip/hsa tracing) over scripts/pmc_unet.py.
#   usage: scripts/pmc_family.sh TAG   -> gpurun_out/TAG/pmc_family.txt
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT=$ROOT/gpurun_out/${1:-pmc_family}; mkdir -p $OUT; export TMPDIR=/tmp
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"
C="GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum"
cd /tmp
for P in a c; do
  CTRS=$A; [ $P = c ] && CTRS=$C
  rm -rf /tmp/pmc_family_$P
  TC_HIPGRAPH=0 timeout -k 5 300 rocprofv3 --kernel-trace --pmc $CTRS -d /tmp/pmc_family_$P -o pmc -- python $ROOT/scripts/pmc_unet.py 2 > $OUT/pmc_family_$P.log 2>&1
done
python $ROOT/scripts/pmc_family.py "$(find /tmp/pmc_family_a -name '*.db' | head -1)" "$(find /tmp/pmc_family_c -name '*.db' | head -1)" > $OUT/pmc_family.txt 2>> $OUT/pmc_family_a.log
cat $OUT/pmc_family.txt
