#!/bin/bash
# usage: scripts/prof_ab.sh ENVVAR val_a val_b   -- rocprofv3 kernel stats of a short bench (6 DDIM steps) per arm
VAR=$1; A=$2; B=$3
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; export TMPDIR=/tmp
bash $ROOT/scripts/which_gpu.sh
cd /tmp
for v in $A $B; do
  rm -rf /tmp/prof_$v
  env $VAR=$v timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o prof -- python $ROOT/bench.py --no-retry --steps 1 --warmup 1 --ddim-steps 6 --no-cpu-baseline --no-roofline > /tmp/prof_$v.log 2>&1
  python $ROOT/scripts/prof_summary.py "$(find /tmp/prof_$v -name '*.db' | head -1)" 30 > $ROOT/gpurun_out/prof_${VAR}_$v.txt 2>&1
done
