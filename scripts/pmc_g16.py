#!/usr/bin/env python3
"""Workload for rocprofv3 --pmc passes over the 160x160 3x3 convolution of UNet level 0 and its timing ablations
(TC_G16_ABLATE: separate kernel instantiations, so the counters come out per variant), three launches each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tooncrafter_amd import ops
hip = ops.backend(); dev = "cuda"; BF = torch.bfloat16
def rnd(*s, scale=1.0): return (torch.randn(*s, device=dev) * scale).to(BF)
os.environ["TC_GEMM_TILE16"] = "2"
frames, h, w_, cin, cout = 32, 40, 64, 320, 320
x, wt, b = rnd(frames * h * w_, cin), rnd(cout, 9 * cin, scale=(9 * cin) ** -0.5), torch.randn(cout, device=dev)
geom = dict(kind="3x3", frames=frames, cin=cin, h_in=h, w_in=w_, h_out=h, w_out=w_, stride=1, upsample=False)
for abl, tall in (("0", "0"), ("2", "0"), ("3", "0"), ("4", "0"), ("0", "2")):
    os.environ["TC_G16_ABLATE"], os.environ["TC_G16_TALL"] = abl, tall
    for _ in range(3):
        hip.gemm(x, wt, b, conv=geom)
    torch.cuda.synchronize()
