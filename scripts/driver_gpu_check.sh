#!/bin/bash
# Re-run the driver's two round-end GPU commands exactly as GPUTEST_rNN.json records them, from clean
# processes; on failure repeat under TC_DEBUG_SYNC=1 (per-launch trace + sync) to name the faulting launch.
mkdir -p gpurun_out/drv
cd "${GRAFT_REPO_ROOT:-.}"
python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > gpurun_out/drv/pytest.log 2>&1; rc1=$?
python3 -c 'import sys; sys.path.insert(0, "."); import __graft_entry__ as e; e.smoke(); print("__SMOKE_OK__")' > gpurun_out/drv/smoke.log 2>&1; rc2=$?
echo "pytest rc=$rc1 smoke rc=$rc2" | tee gpurun_out/drv/rc.txt
tail -3 gpurun_out/drv/pytest.log; tail -3 gpurun_out/drv/smoke.log
if [ $rc2 -ne 0 ]; then
  TC_DEBUG_SYNC=1 AMD_SERIALIZE_KERNEL=3 python3 -c 'import sys; sys.path.insert(0, "."); import __graft_entry__ as e; e.smoke()' > gpurun_out/drv/smoke_dbg.log 2>&1
  echo "smoke dbg rc=$?"; tail -5 gpurun_out/drv/smoke_dbg.log
fi
if [ $rc1 -ne 0 ]; then
  TC_DEBUG_SYNC=1 AMD_SERIALIZE_KERNEL=3 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > gpurun_out/drv/pytest_dbg.log 2>&1
  echo "pytest dbg rc=$?"; grep -n "^\[tc\]" gpurun_out/drv/pytest_dbg.log | tail -3
fi
