#!/bin/bash
# usage: scripts/pmc_run.sh <tag> <script.py> [kernel-name-filter]   -> gpurun_out/pmc_<tag>_{a,b}.json
# Two counter passes (8 SQ slots each), --kernel-trace only (never with sys/hip/hsa tracing: gpurun refuses that).
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
TAG=$1; SCRIPT=$2; FLT=${3:-}
OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"
B="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM"
cd /tmp
for P in a b; do
  CTRS=$A; [ $P = b ] && CTRS=$B
  rm -rf /tmp/pmc_${TAG}_$P
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $CTRS -d /tmp/pmc_${TAG}_$P -o pmc -- python $ROOT/$SCRIPT > $OUT/pmc_${TAG}_$P.log 2>&1
  python $ROOT/scripts/pmc_dump.py "$(find /tmp/pmc_${TAG}_$P -name '*.db' | head -1)" "$FLT" > $OUT/pmc_${TAG}_$P.json 2>> $OUT/pmc_${TAG}_$P.log
done
