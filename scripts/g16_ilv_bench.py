#!/usr/bin/env python3
"""The 160x160 GEMM's K loop: plain (TC_G16_ILV=0) against the two loops with the tile requests between the MFMAs
(1, 2; csrc/gemm16.hip), interleaved in one process, on the UNet's convolution and long-K linear shapes at B = 2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tooncrafter_amd import ops
dev, BF = "cuda", torch.bfloat16
hip = ops.backend()

def timeit(fn, iters=20, reps=3):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / iters)
    return min(ts)

ARMS = [("plain", 0, 0), ("loop2", 2, 0), ("tall", 0, 2), ("tall+loop1", 1, 2), ("tall+loop2", 2, 2)]

def ab(fn, flops, tag):
    r = {}
    for _ in range(2):
        for name, ilv, tall in ARMS:
            os.environ["TC_G16_ILV"], os.environ["TC_G16_TALL"] = str(ilv), str(tall)
            r.setdefault(name, []).append(timeit(fn))
    os.environ["TC_G16_ILV"] = os.environ["TC_G16_TALL"] = "0"
    t = {v: min(x) * 1e3 for v, x in r.items()}
    print(f"{tag:34s} plain {t['plain']:7.1f} us {flops / t['plain'] / 1e6:7.1f} TF/s | " +
          " | ".join(f"{n} {t[n]:7.1f} x{t['plain'] / t[n]:5.3f}" for n, _, _ in ARMS[1:]), flush=True)

def conv(frames, h, w, cin, cout, tag, t3=False):
    x = torch.randn(frames * h * w, cin, device=dev).to(BF)
    taps = 3 if t3 else 9
    wt = (torch.randn(cout, taps * cin, device=dev) * (taps * cin) ** -0.5).to(BF); b = torch.randn(cout, device=dev)
    res = torch.randn(frames * h * w, cout, device=dev).to(BF)
    geom = dict(kind="t3", frames=frames, t_len=16, cin=cin, h_out=h, w_out=w) if t3 else \
        dict(kind="3x3", frames=frames, cin=cin, h_in=h, w_in=w, h_out=h, w_out=w, stride=1, upsample=False)
    ab(lambda: hip.gemm(x, wt, b, conv=geom, residual=res), 2.0 * frames * h * w * cout * taps * cin,
       f"{'convT3' if t3 else 'conv3x3'} {tag} {cin}->{cout}")

def lin(m, n, k, tag):
    a = torch.randn(m, k, device=dev).to(BF); w = (torch.randn(n, k, device=dev) * k ** -0.5).to(BF)
    b = torch.randn(n, device=dev); r = torch.randn(m, n, device=dev).to(BF)
    ab(lambda: hip.gemm(a, w, b, residual=r), 2.0 * m * n * k, f"linear {tag} {m}x{n}x{k}")

os.environ["TC_GEMM_TILE16"] = "2"; os.environ["TC_GEMM8"] = "0"; os.environ["TC_GEMM_WS"] = "0"
conv(32, 40, 64, 320, 320, "L0"); conv(32, 40, 64, 640, 320, "L0"); conv(32, 40, 64, 960, 320, "L0")
conv(32, 20, 32, 640, 640, "L1"); conv(32, 20, 32, 1280, 640, "L1"); conv(32, 10, 16, 1280, 1280, "L2")
conv(32, 40, 64, 320, 320, "L0", t3=True); conv(32, 20, 32, 640, 640, "L1", t3=True); conv(32, 10, 16, 1280, 1280, "L2", t3=True)
lin(81920, 320, 1280, "L0 ff2"); lin(20480, 640, 2560, "L1 ff2"); lin(81920, 960, 320, "L0 qkv"); lin(20480, 1920, 640, "L1 qkv")
lin(5120, 1280, 5120, "L2 ff2"); lin(8192, 8000, 8192, "square-ish")
