#!/bin/bash
# Round 6, call 6: GroupNorm's finalize inside the apply blocks' prologue (TC_GN_FINALIZE=apply): tests + forward A/B + decoder A/B.
cd "$(dirname "$0")/.."
TAG=${1:-r6c6}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -p no:cacheprovider -k "groupnorm" -s > $OUT/pytest_gn.log 2>&1; echo "pytest gn rc=$?" | tee -a $OUT/log.txt
grep -E "finalize in apply vs|passed|failed" $OUT/pytest_gn.log | head -12
timeout 600 python scripts/forward_env_ab.py TC_GN_FINALIZE launch apply > $OUT/forward_ab_gn_fin.txt 2> $OUT/ab.err; echo "forward ab rc=$?" | tee -a $OUT/log.txt
tail -7 $OUT/forward_ab_gn_fin.txt
for v in launch apply launch apply; do TC_GN_FINALIZE=$v timeout 300 python scripts/forward_breakdown.py --decoder 2>/dev/null | head -2 | tail -1 | cut -c1-200 | sed "s/^/$v: /" | tee -a $OUT/decoder_ab.txt; done
