#!/bin/bash
# Diagnose the visible GPU: plain-PyTorch health check first, then our smoke test (traced on failure).
mkdir -p gpurun_out/diag; export TMPDIR=/tmp
N=$(bash scripts/which_gpu.sh | tee gpurun_out/diag/which_gpu.txt | sed -n 's/visible kfd node \([0-9]*\).*/\1/p' | head -1)
timeout 300 python scripts/gpu_health.py > gpurun_out/diag/health_node$N.txt 2>&1; echo "health rc=$? (node $N)" | tee -a gpurun_out/diag/health_node$N.txt
timeout 600 python -c 'import __graft_entry__ as e; e.smoke()' > gpurun_out/diag/smoke_node$N.txt 2>&1; rc=$?; echo "smoke rc=$rc (node $N)" | tee -a gpurun_out/diag/smoke_node$N.txt
if [ $rc -ne 0 ]; then
  TC_DEBUG_SYNC=1 AMD_SERIALIZE_KERNEL=3 timeout 600 python -c 'import __graft_entry__ as e; e.smoke()' > gpurun_out/diag/smoke_traced_node$N.txt 2>&1
  echo "traced smoke rc=$? (node $N)"; grep "^\[tc\]" gpurun_out/diag/smoke_traced_node$N.txt | tail -2 | cut -c1-600; grep -v "^\[tc\]" gpurun_out/diag/smoke_traced_node$N.txt | tail -4
fi
tail -3 gpurun_out/diag/health_node$N.txt
