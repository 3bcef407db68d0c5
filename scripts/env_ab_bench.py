#!/usr/bin/env python3
"""usage: env_ab_bench.py VAR valueA valueB [--big]  -- interleaved in-process A/B of one tuning switch that the
library reads per call (TC_GEMM_EPI lds|direct, TC_GEMM_TILE16 0|2) on the GEMM shapes of a B=2 UNet forward."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tooncrafter_amd import ops
from tooncrafter_amd._lib import ACT_GEGLU, ACT_NONE, ACT_SILU
VAR, VA, VB = sys.argv[1:4]
dev, BF = "cuda", torch.bfloat16
hip = ops.backend()

def timeit(fn, iters=20, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / iters)
    return sorted(ts)[len(ts) // 2]

def ab(fn, flops, tag):
    r = {}
    for _ in range(2):
        for v in (VA, VB):
            os.environ[VAR] = v
            r.setdefault(v, []).append(timeit(fn))
    ta, tb = min(r[VA]), min(r[VB])
    print(f"{tag:40s} {VA:>6s} {ta*1e3:8.1f} us {flops/ta/1e9:7.1f} TF/s | {VB:>6s} {tb*1e3:8.1f} us {flops/tb/1e9:7.1f} TF/s | x{ta/tb:5.2f}", flush=True)

def lin(m, n, k, tag, act=ACT_NONE, res=True, rb=False):
    a = torch.randn(m, k, device=dev).to(BF); w = (torch.randn(n, k, device=dev) * k ** -0.5).to(BF)
    b = torch.randn(n, device=dev); r = torch.randn(m, n, device=dev).to(BF) if res and act != ACT_GEGLU else None
    ab(lambda: hip.gemm(a, w, b, act=act, residual=r), 2.0 * m * n * k, f"linear {tag} {m}x{n}x{k}")

def conv(frames, h, w, cin, cout, tag, t3=False):
    x = torch.randn(frames * h * w, cin, device=dev).to(BF)
    taps = 3 if t3 else 9
    wt = (torch.randn(cout, taps * cin, device=dev) * (taps * cin) ** -0.5).to(BF); b = torch.randn(cout, device=dev)
    rb = torch.randn(2, cout, device=dev)
    geom = dict(kind="t3", frames=frames, t_len=16, cin=cin, h_out=h, w_out=w) if t3 else \
        dict(kind="3x3", frames=frames, cin=cin, h_in=h, w_in=w, h_out=h, w_out=w, stride=1, upsample=False)
    ab(lambda: hip.gemm(x, wt, b, conv=geom, row_bias=rb, row_div=frames * h * w // 2), 2.0 * frames * h * w * cout * taps * cin,
       f"{'convT3' if t3 else 'conv3x3'} {tag} {cin}->{cout}")

lin(81920, 320, 320, "L0 proj"); lin(81920, 960, 320, "L0 qkv", res=False); lin(81920, 2560, 320, "L0 geglu", act=ACT_GEGLU)
lin(81920, 320, 1280, "L0 ff2")
lin(20480, 640, 640, "L1 proj"); lin(20480, 1920, 640, "L1 qkv", res=False); lin(20480, 5120, 640, "L1 geglu", act=ACT_GEGLU)
lin(20480, 640, 2560, "L1 ff2")
lin(5120, 1280, 1280, "L2 proj"); lin(5120, 3840, 1280, "L2 qkv", res=False); lin(5120, 1280, 5120, "L2 ff2")
lin(1280, 1280, 1280, "L3 proj")
conv(32, 10, 16, 1280, 1280, "L2"); conv(32, 10, 16, 1280, 1280, "L2", t3=True); conv(32, 5, 8, 1280, 1280, "L3", t3=True)
if "--big" in sys.argv:
    conv(32, 40, 64, 320, 320, "L0"); conv(32, 20, 32, 640, 640, "L1"); lin(4096, 4096, 4096, "square 4k", res=False)
