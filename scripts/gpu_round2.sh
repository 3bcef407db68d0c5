#!/bin/bash
# Round-2 GPU visit.  usage: scripts/gpu_round2.sh stage...   (stages: drv noise bench benchbd profdec pmcattn probe gemmbench)
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for s in "$@"; do
  t0=$(date +%s)
  case $s in
    drv)     bash scripts/driver_gpu_check.sh > $OUT/drv_stage.log 2>&1 ;;
    full)    timeout 1500 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -p no:cacheprovider > $OUT/fullsize.log 2>&1 ;;
    noise)   timeout 1200 python scripts/noise_floor.py > $OUT/noise_floor.txt 2> $OUT/noise_floor.err ;;
    bench)   timeout 1500 python bench.py --steps 2 --warmup 1 > $OUT/bench_full.log 2>&1 ;;
    benchq)  timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_quick.log 2>&1 ;;
    benchbd) timeout 1500 python bench.py --steps 1 --warmup 1 --batched-decode 4 --no-cpu-baseline --no-roofline > $OUT/bench_batched_decode.log 2>&1 ;;
    profdec) (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/profdec -o prof -- python $OLDPWD/bench.py --no-retry --steps 1 --warmup 1 --ddim-steps 1 --no-cpu-baseline --no-roofline > $OLDPWD/$OUT/profdec.log 2>&1; python $OLDPWD/scripts/prof_summary.py "$(find /tmp/profdec -name '*.db' | head -1)" 40 > $OLDPWD/$OUT/profdec_stats.txt 2>> $OLDPWD/$OUT/profdec.log) ;;
    prof)    (cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats -d /tmp/profclip -o prof -- python $OLDPWD/bench.py --no-retry --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $OLDPWD/$OUT/prof.log 2>&1; python $OLDPWD/scripts/prof_summary.py "$(find /tmp/profclip -name '*.db' | head -1)" 40 > $OLDPWD/$OUT/prof_stats.txt 2>> $OLDPWD/$OUT/prof.log) ;;
    pmcattn) bash scripts/pmc_run.sh attn scripts/pmc_attn.py attn ;;
    probe)   (cd /tmp && hipcc --offload-arch=gfx950 -O2 $OLDPWD/scripts/probes/tr_probe.hip -o /tmp/tr_probe 2>/dev/null && /tmp/tr_probe) > $OUT/tr_probe.txt 2>&1 ;;
    fp8)     timeout 900 python -m pytest tests/test_gpu_fp8.py tests/test_gpu_fullsize.py -m gpu -q -s -p no:cacheprovider -k "fp8 or mx or quantiser or ineligible" > $OUT/fp8_tests.log 2>&1 ;;
    benchfp8) timeout 900 python bench.py --fp8 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/bench_fp8.log 2>&1
              timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/bench_bf16_same_box.log 2>&1 ;;
    proffp8) (cd /tmp && TC_FP8=1 timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/proffp8 -o prof -- python $OLDPWD/bench.py --fp8 --no-retry --steps 1 --warmup 0 --ddim-steps 6 --no-cpu-baseline --no-roofline > $OLDPWD/$OUT/proffp8.log 2>&1; python $OLDPWD/scripts/prof_summary.py "$(find /tmp/proffp8 -name '*.db' | head -1)" 30 > $OLDPWD/$OUT/proffp8_stats.txt 2>> $OLDPWD/$OUT/proffp8.log) ;;
    torchrun1) timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/bench_torchrun1.log 2>&1 ;;
    mxbench) timeout 300 python scripts/mx_bench.py > $OUT/mx_bench.txt 2>&1 ;;
    gemmbench) timeout 600 python scripts/gemm_bench.py > $OUT/gemm_bench.log 2>&1 ;;
    attncheck) timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_guard.py tests/test_gpu_properties.py -m gpu -q -p no:cacheprovider -k "attention" > $OUT/attn_check.log 2>&1
               if [ $? -ne 0 ]; then echo "DMA attention FAILED parity: falling back to TC_ATTN_STAGE=reg for the rest of this visit" | tee -a $OUT/round2.log; export TC_ATTN_STAGE=reg; fi ;;
    attnab)  (for v in reg dma reg dma; do echo "== TC_ATTN_STAGE=$v"; TC_ATTN_STAGE=$v timeout 300 python scripts/gemm_bench.py --attn-only 2>&1 | grep -v amdgpu.ids; done) > $OUT/attn_ab.log 2>&1 ;;
    attnbench) timeout 600 python scripts/attn_bench.py > $OUT/attn_bench.log 2>&1 ;;
    *) echo "unknown stage $s" ;;
  esac
  echo "stage $s rc=$? $(( $(date +%s) - t0 ))s" | tee -a $OUT/round2.log
done
