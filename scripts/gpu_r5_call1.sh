#!/bin/bash
# Round 5, first GPU visit: the kernels of round 4's GPU-less session.  Gated tests first; the A/Bs only if they pass.
cd "$(dirname "$0")/.."
TAG=${1:-r5c1}
bash scripts/gpu_round5.sh $TAG unver
if tail -3 gpurun_out/$TAG/pytest_unverified.log | grep -q "passed" && ! tail -3 gpurun_out/$TAG/pytest_unverified.log | grep -q "failed\|error"; then
  bash scripts/gpu_round5.sh $TAG halo gn fuseunet fuseclip
else
  bash scripts/gpu_round5.sh $TAG halodbg gn
fi
tail -30 gpurun_out/$TAG/pytest_unverified.log
