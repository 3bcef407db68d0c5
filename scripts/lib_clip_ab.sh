#!/bin/bash
# whole-clip A/B of two BUILDS of the library on one box (scripts/clip_ab2.sh's output format)
cp tooncrafter_amd/libtooncrafter_hip.so /tmp/new.so
for r in 1 2; do for v in new old; do
  if [ $v = old ]; then cp scripts/bin/prev/libtooncrafter_hip.so tooncrafter_amd/libtooncrafter_hip.so; else cp /tmp/new.so tooncrafter_amd/libtooncrafter_hip.so; fi
  timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/ab.log 2>&1
  python - "[$v]" <<'PY'
import json, sys
line = [l for l in open("/tmp/ab.log") if l.startswith("{")]
if not line:
    print(sys.argv[1], "FAILED:", open("/tmp/ab.log").read()[-300:])
else:
    d = json.loads(line[0])
    print(sys.argv[1], "frames/s", d["value"], "ms/clip", d["ms_per_step"], "gemm ms/fwd", d["roofline"]["unet"]["gemm_ms_per_unet_fwd_b2"],
          "frac", d["roofline"]["frac"], d.get("boundary_host_overhead"), "GN ms/fwd", d["roofline_hbm"]["ms_per_unet_fwd_b2"])
PY
done; done
cp /tmp/new.so tooncrafter_amd/libtooncrafter_hip.so
