#!/usr/bin/env python3
"""Which hipBLASLt kernels the library picks for the yardstick shapes of profiles/r03_ws_bench_hipblaslt_yardstick.txt (run under
`rocprofv3 --kernel-trace --stats`: the Tensile kernel names encode macro tile, depth-U, wave tiling, LDS and prefetch options;
the code objects themselves are disassembled on the build box).  Script only: torch.nn.functional.linear is not the product."""
import torch
import torch.nn.functional as F
dev, BF = "cuda", torch.bfloat16
for m, n, k in ((8192, 8192, 8192), (4096, 4096, 4096), (5120, 10240, 1280), (20480, 5120, 640), (20480, 640, 2560),
                (5120, 1280, 5120), (81920, 320, 2880), (20480, 640, 5760)):
    a = torch.randn(m, k, device=dev).to(BF)
    w = torch.randn(n, k, device=dev).to(BF)
    for _ in range(3):
        F.linear(a, w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        F.linear(a, w)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"hipBLASLt {m}x{n}x{k}: {ms*1e3:8.1f} us {2.0*m*n*k/ms/1e9:7.1f} TF/s", flush=True)
