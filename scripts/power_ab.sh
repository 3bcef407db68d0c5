#!/bin/bash
# usage: scripts/power_ab.sh ENVVAR a b  -- bench.py per arm with a 100-ms rocm-smi sampler (sclk, power) beside it
VAR=$1; A=$2; B=$3
bash scripts/which_gpu.sh
timeout 300 python scripts/gpu_health.py 2>&1 | tail -2
env $VAR=$A timeout 600 python bench.py --steps 1 --warmup 1 --ddim-steps 10 --no-cpu-baseline --no-roofline > /tmp/ab0.log 2>&1; echo "no-sampler run rc=$?"; grep -v "^{" /tmp/ab0.log | tail -3
for v in $A $B; do
  ( while true; do rocm-smi --showclocks --showpower --json 2>/dev/null | tr -d '\n'; echo; sleep 0.1; done ) > /tmp/smi_$v.log &
  SMI=$!
  env $VAR=$v timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > /tmp/ab.log 2>&1
  kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
  python - "$VAR=$v" /tmp/smi_$v.log <<'PY'
import json, re, sys
line = [l for l in open("/tmp/ab.log") if l.startswith("{")]
d = json.loads(line[0]) if line else None
if not d:
    print("FAILED:", open("/tmp/ab.log").read()[-400:])
sclk, pw = [], []
for l in open(sys.argv[2]):
    try:
        j = json.loads(l)
    except Exception:
        continue
    for card in j.values():
        for k, v in card.items():
            if "sclk" in k.lower() and "mhz" in str(v).lower():
                m = re.search(r"(\d+)\s*mhz", str(v).lower());  sclk.append(int(m.group(1))) if m else None
            if "power" in k.lower() and "(w)" in k.lower():
                try: pw.append(float(v))
                except Exception: pass
busy = [s for s, p in zip(sclk, pw) if p > 600] if len(sclk) == len(pw) else sclk
print(sys.argv[1], "frames/s", d["value"] if d else "FAILED", "sampler ms", d["stage_ms_per_clip"]["ddim_sampler"] if d else None,
      "| samples", len(sclk), "sclk under load avg %.0f MHz min %d max %d" % (sum(busy) / max(len(busy), 1), min(busy or [0]), max(busy or [0])),
      "| power under load avg %.0f W max %.0f" % (sum(p for p in pw if p > 600) / max(len([p for p in pw if p > 600]), 1), max(pw or [0])))
PY
done
rocm-smi --showclocks --showpower --json 2>/dev/null | head -c 1500
