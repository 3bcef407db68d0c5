#!/bin/bash
# Build the library with another GELU formulation (TC_GELU_VARIANT: 0 = correctly rounded division + copysign form,
# 2 = the same form on v_rcp_f32; the default is the max(x,0) - |x| u form) into scripts/bin/gelu<N>/ for an A/B of the
# GEGLU epilogues with scripts/bin/gemm8_bench (LD_LIBRARY_PATH=scripts/bin/gelu<N>).
set -e
cd "$(dirname "$0")/.."
for v in "$@"; do
  out=scripts/bin/gelu$v; mkdir -p $out/o
  for f in gemm gemm_wide gemm16 gemm8 ff_fused gemm_ws gemm_mx attention norm elementwise; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Iinclude -Itooncrafter_amd/csrc -DTC_GELU_VARIANT=$v \
      -DTC_SRC_DIGEST='"ab"' -c tooncrafter_amd/csrc/$f.hip -o $out/o/$f.o &
  done
  wait
  hipcc --offload-arch=gfx950 -shared -fPIC $out/o/*.o -o $out/libtooncrafter_hip.so
  rm -rf $out/o
done
