#!/bin/bash
# usage: scripts/clip_ab2.sh "ENV_A" "ENV_B" [rounds]  -- whole-clip A/B on ONE box: bench.py --steps 2 per arm, interleaved;
# each arm is a string of VAR=value assignments (may be empty = the defaults)
A=$1; B=$2; R=${3:-2}
for r in $(seq $R); do for v in "$A" "$B"; do
  env $v timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/ab.log 2>&1
  python - "[$v]" <<'PY'
import json, sys
line = [l for l in open("/tmp/ab.log") if l.startswith("{")]
if not line:
    print(sys.argv[1], "FAILED:", open("/tmp/ab.log").read()[-300:])
else:
    d = json.loads(line[0])
    print(sys.argv[1], "frames/s", d["value"], "ms/clip", d["ms_per_step"], d["stage_ms_per_clip"], "gemm ms/fwd",
          d["roofline"]["unet"]["gemm_ms_per_unet_fwd_b2"], "frac", d["roofline"]["frac"], d.get("boundary_host_overhead"),
          "GN ms/fwd", d["roofline_hbm"]["ms_per_unet_fwd_b2"])
PY
done; done
