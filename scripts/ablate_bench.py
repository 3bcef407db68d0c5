#!/usr/bin/env python3
"""Where does a short-K tile's time go?  Times the 128x128 kernel of csrc/gemm.hip in its ablation builds
(scripts/ablate_gemm.sh: TC_ABLATE bitmask 1 no MFMAs | 2 no steady-state tile loads | 4 no epilogue | 8 GEGLU without erf |
16 no global stores) against the product build, one subprocess per library, two interleaved rounds.

    bash scripts/ablate_gemm.sh 1 2 4 8 16 3 && python scripts/ablate_bench.py > gpurun_out/ablate_bench.txt
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = [("product", "-"), ("no MFMAs (1)", "1"), ("no steady-state loads (2)", "2"), ("neither (3)", "3"),
            ("no epilogue (4)", "4"), ("GEGLU without erf (8)", "8"), ("no global stores (16)", "16")]
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch
    import tooncrafter_amd._lib as L
    if sys.argv[2] != "-":
        L.LIB_PATH = os.path.join(ROOT, "tooncrafter_amd", "build", "ablate", f"libtooncrafter_hip_ab{sys.argv[2]}.so")
    from tooncrafter_amd import ops
    from tooncrafter_amd._lib import ACT_GEGLU, ACT_NONE
    os.environ["TC_GEMM_WS"] = "0"
    os.environ["TC_GEMM_WIDE"] = "0"
    hip = ops.backend()
    BF = torch.bfloat16
    out = []
    for tag, m, n, k, geglu, res in [("L0 GEGLU", 81920, 2560, 320, True, False), ("L0 qkv", 81920, 960, 320, False, False),
                                     ("L0 ff2+res", 81920, 320, 1280, False, True), ("L1 GEGLU", 20480, 5120, 640, True, False),
                                     ("L1 proj+res", 20480, 640, 640, False, True), ("L2 GEGLU 128-tile", 5120, 10240, 1280, True, False),
                                     ("L2 proj+res", 5120, 1280, 1280, False, True)]:
        a = [torch.randn(m, k, device="cuda").to(BF) for _ in range(4)]
        w = (torch.randn(n, k, device="cuda") * k ** -0.5).to(BF)
        b = torch.randn(n, device="cuda")
        n_out = n // 2 if geglu else n
        r = [torch.randn(m, n_out, device="cuda").to(BF) for _ in range(4)] if res else [None] * 4
        o = torch.empty(m, n_out, device="cuda", dtype=BF)
        act = ACT_GEGLU if geglu else ACT_NONE
        for i in range(4):
            hip.gemm(a[i], w, b, act=act, residual=r[i], out=o)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(16):
                hip.gemm(a[i % 4], w, b, act=act, residual=r[i % 4], out=o)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 16 * 1e3)
        out.append(f"{tag} {sorted(ts)[1]:7.1f}")
    print(" | ".join(out), flush=True)
    sys.exit(0)
for rnd in range(2):
    for name, tag in VARIANTS:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", tag], capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if "|" in l]
        print(f"{name:28s} {line[-1] if line else 'FAILED ' + p.stderr[-300:]}", flush=True)
