#!/bin/bash
# Does a concurrent rocm-smi poller change the clip time?  Same box, alternating: no poller / poller at 100 ms / poller at 1 s
bash scripts/which_gpu.sh
run() {  # $1 = label
  timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > /tmp/ab.log 2>&1
  python - "$1" <<'PY'
import json, sys
line = [l for l in open("/tmp/ab.log") if l.startswith("{")]
if not line: print(sys.argv[1], "FAILED", open("/tmp/ab.log").read()[-300:])
else:
    d = json.loads(line[0]); print(sys.argv[1], "frames/s", d["value"], "ms/clip", d["ms_per_step"], d["stage_ms_per_clip"])
PY
}
run "no poller          "
( while true; do rocm-smi --showclocks --showpower --json > /dev/null 2>&1; sleep 0.1; done ) & P=$!
run "poller 100 ms      "; kill $P; wait $P 2>/dev/null
run "no poller          "
( while true; do rocm-smi --showclocks --json > /dev/null 2>&1; sleep 1; done ) & P=$!
run "poller 1 s         "; kill $P; wait $P 2>/dev/null
( while true; do cat /sys/class/drm/card*/device/pp_dpm_sclk > /dev/null 2>&1; sleep 0.05; done ) & P=$!
run "sysfs pp_dpm_sclk 50ms"; kill $P; wait $P 2>/dev/null
cat /sys/class/drm/card*/device/power_dpm_force_performance_level 2>&1 | head -3
cat /sys/class/drm/card*/device/pp_dpm_sclk 2>&1 | head -5
