#!/bin/bash
# usage: scripts/clip_ab.sh ENVVAR val_a val_b [rounds]   -- whole-clip A/B on ONE box: bench.py --steps 2 per arm, interleaved
VAR=$1; A=$2; B=$3; R=${4:-2}
bash scripts/which_gpu.sh
for r in $(seq $R); do for v in $A $B; do
  env $VAR=$v timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/ab.log 2>&1
  python - "$VAR=$v" <<'PY'
import json, sys
line = [l for l in open("/tmp/ab.log") if l.startswith("{")]
if not line:
    print(sys.argv[1], "FAILED:", open("/tmp/ab.log").read()[-300:])
else:
    d = json.loads(line[0])
    print(sys.argv[1], "frames/s", d["value"], "ms/clip", d["ms_per_step"], d["stage_ms_per_clip"], "gemm ms/fwd",
          d["roofline"]["gemm_ms_per_unet_fwd_b2"], "frac", d["roofline"]["frac"], d.get("boundary_host_overhead"))
PY
done; done
