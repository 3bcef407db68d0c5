#!/bin/bash
# Round 6, evidence run on the final tree: the driver's commands, the per-kernel trace with its own lease calibration, the counter
# passes (per family + fabric traffic), the configs[3] / configs[4] bench lines, the per-operator breakdowns.
cd "$(dirname "$0")/.."
TAG=${1:-r6final}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
REPO=$(pwd)
t0=$(date +%s); timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest_all.log 2>&1; echo "pytest rc=$? $(( $(date +%s) - t0 )) s" | tee -a $OUT/log.txt
tail -3 $OUT/pytest_all.log
cp gpurun_out/ddim50_parity_*.txt gpurun_out/encoder_fullsize_parity.txt $OUT/ 2>/dev/null
t0=$(date +%s); timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$? $(( $(date +%s) - t0 )) s" | tee -a $OUT/log.txt
tail -2 $OUT/smoke.log
t0=$(date +%s); timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? $(( $(date +%s) - t0 )) s" | tee -a $OUT/log.txt
head -c 300 $OUT/bench.json; echo
(cd /tmp && rm -rf /tmp/profclip && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/profclip -o prof -- python $REPO/bench.py --steps 1 --warmup 0 --ddim-steps 10 --no-cpu-baseline --no-roofline --no-extras > $REPO/$OUT/prof.log 2>&1; python $REPO/scripts/prof_summary.py "$(find /tmp/profclip -name '*.db' | head -1)" 60 $REPO/$OUT/prof.log 10 > $REPO/$OUT/prof_stats.txt 2>> $REPO/$OUT/prof.log); echo "prof rc=$?" | tee -a $OUT/log.txt
head -4 $OUT/prof_stats.txt | cut -c1-200; grep "GEMM family" $OUT/prof_stats.txt | cut -c1-250
bash scripts/pmc_family.sh $TAG > $OUT/pmc_family.out 2>&1; echo "pmc family rc=$?" | tee -a $OUT/log.txt
bash scripts/pmc_traffic.sh $TAG 2 > $OUT/pmc_traffic.out 2>&1; echo "pmc traffic rc=$?" | tee -a $OUT/log.txt
tail -2 $OUT/pmc_traffic.out
timeout 600 python scripts/forward_breakdown.py > $OUT/forward_breakdown_unet.txt 2> $OUT/forward_breakdown.err; echo "breakdown rc=$?" | tee -a $OUT/log.txt
timeout 600 python scripts/forward_breakdown.py --decoder > $OUT/forward_breakdown_decoder.txt 2>> $OUT/forward_breakdown.err; echo "breakdown dec rc=$?" | tee -a $OUT/log.txt
timeout 600 python bench.py --fp8 --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench_fp8.json 2> $OUT/bench_fp8.err; echo "bench fp8 rc=$?" | tee -a $OUT/log.txt
head -c 200 $OUT/bench_fp8.json; echo
timeout 600 python bench.py --batched-decode 2 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench_b2.json 2> $OUT/bench_b2.err; echo "bench b2 rc=$?" | tee -a $OUT/log.txt
head -c 200 $OUT/bench_b2.json; echo
