#!/bin/bash
# Ablation builds of csrc/gemm.hip (compile-time TC_ABLATE bitmask; the product build defines nothing):
#   1 no MFMAs | 2 no steady-state tile loads | 4 no epilogue | 8 GEGLU without erf | 16 no global stores
# -> tooncrafter_amd/build/ablate/libtooncrafter_hip_ab<N>.so (the other objects are the product build's).
set -eu
cd "$(dirname "$0")/.."
python -c "from tooncrafter_amd import build; build.build(verbose=False)"
D=tooncrafter_amd/build; mkdir -p $D/ablate
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Iinclude -Itooncrafter_amd/csrc \
    -DTC_SRC_DIGEST="\"ablate$n\"" -DTC_ABLATE=$n -c tooncrafter_amd/csrc/gemm.hip -o $D/ablate/gemm_ab$n.o &
done
wait
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $D/ablate/gemm_ab$n.o $D/gemm_wide.o $D/gemm16.o $D/gemm_ws.o $D/gemm_mx.o \
    $D/attention.o $D/norm.o $D/elementwise.o -o $D/ablate/libtooncrafter_hip_ab$n.so
done
ls -la $D/ablate/*.so
