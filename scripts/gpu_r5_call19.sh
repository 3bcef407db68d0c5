#!/bin/bash
# Round 5, GPU visit 19: the engine clock in situ (scripts/clock_insitu.py) with a 100-ms rocm-smi sampler beside it.
cd "$(dirname "$0")/.."
TAG=${1:-r5c19}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
bash scripts/which_gpu.sh 2>/dev/null | tail -3
( while true; do date +%s.%N | tr '\n' ' '; rocm-smi --showclocks --showpower --json 2>/dev/null | tr -d '\n'; echo; sleep 0.1; done ) > $OUT/smi.log &
SMI=$!
timeout 420 python scripts/clock_insitu.py > $OUT/clock_insitu.txt 2> $OUT/clock_insitu.err; echo "clock_insitu rc=$?" | tee -a $OUT/log.txt
kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
cat $OUT/clock_insitu.txt
tail -5 $OUT/clock_insitu.err
python - $OUT/smi.log <<'PY'
import json, re, sys
rows = []
for l in open(sys.argv[1]):
    try:
        t, js = l.split(" ", 1)
        j = json.loads(js)
    except Exception:
        continue
    sclk = pw = None
    for card in j.values():
        for k, v in card.items():
            if "sclk clock speed" in k.lower():
                m = re.search(r"(\d+)\s*mhz", str(v).lower()); sclk = int(m.group(1)) if m else None
            if "power" in k.lower() and "(w)" in k.lower():
                try: pw = float(v)
                except Exception: pass
    rows.append((float(t), sclk, pw))
busy = [r for r in rows if r[2] and r[2] > 500]
print(f"rocm-smi: {len(rows)} samples, {len(busy)} above 500 W")
if busy:
    print("  under load: sclk avg %.0f min %d max %d MHz | power avg %.0f max %.0f W" % (
        sum(r[1] for r in busy) / len(busy), min(r[1] for r in busy), max(r[1] for r in busy),
        sum(r[2] for r in busy) / len(busy), max(r[2] for r in busy)))
PY
