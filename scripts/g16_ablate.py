#!/usr/bin/env python3
"""Where the time of the 160x160 3x3 convolution goes: TC_G16_ABLATE timing builds (csrc/gemm16.hip), interleaved in one
process.  0 = the product kernel; 1 = A requested for one of the three dx taps (the traffic of a dx-shared slab);
2 = no A requests; 3 = no W requests; 4 = no requests (MFMAs + fragment reads + epilogue).  Results of 1..4 are wrong."""
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

def conv(frames, h, w, cin, cout, tag):
    x = torch.randn(frames * h * w, cin, device=dev).to(BF)
    wt = (torch.randn(cout, 9 * cin, device=dev) * (9 * cin) ** -0.5).to(BF); b = torch.randn(cout, device=dev)
    geom = dict(kind="3x3", frames=frames, cin=cin, h_in=h, w_in=w, h_out=h, w_out=w, stride=1, upsample=False)
    fl = 2.0 * frames * h * w * cout * 9 * cin
    r = {}
    for _ in range(2):
        for a in ("0", "1", "2", "3", "4"):
            os.environ["TC_G16_ABLATE"] = a
            r.setdefault(a, []).append(timeit(lambda: hip.gemm(x, wt, b, conv=geom)))
    os.environ["TC_G16_ABLATE"] = "0"
    t = {a: min(v) * 1e3 for a, v in r.items()}
    print(f"{tag:28s} full {t['0']:7.1f} us {fl / t['0'] / 1e6:7.1f} TF/s | A on 1 of 3 taps {t['1']:7.1f} | no A {t['2']:7.1f} | "
          f"no W {t['3']:7.1f} | no requests {t['4']:7.1f}", flush=True)

os.environ["TC_GEMM_TILE16"] = "2"
conv(32, 40, 64, 320, 320, "conv3x3 L0 320->320")
conv(32, 40, 64, 640, 320, "conv3x3 L0 640->320")
conv(32, 20, 32, 640, 640, "conv3x3 L1 640->640")
conv(32, 20, 32, 1280, 640, "conv3x3 L1 1280->640")
conv(32, 10, 16, 1280, 1280, "conv3x3 L2 1280->1280")
