#!/bin/bash
# HBM-side traffic of the GEMM family and of GroupNorm in a B=2 UNet forward, for bench.py's roofline.traffic /
# roofline_hbm.traffic: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes
# (MI355X_MICROARCH.md, HBM section; never together with sys/hip/hsa tracing) over scripts/pmc_unet.py.
#   usage: scripts/pmc_traffic.sh TAG [forwards]   -> gpurun_out/TAG/r06_pmc_{unet,gn}_traffic.json
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
TAG=$1; N=${2:-2}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_traffic_$C
  TC_HIPGRAPH=0 timeout -k 5 400 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_traffic_$C -o pmc -- python $ROOT/scripts/pmc_unet.py $N > $OUT/pmc_traffic_$C.log 2>&1
done
python $ROOT/scripts/pmc_traffic.py "$(find /tmp/pmc_traffic_FETCH_SIZE -name '*.db' | head -1)" "$(find /tmp/pmc_traffic_WRITE_SIZE -name '*.db' | head -1)" $N $OUT
