#!/usr/bin/env python3
"""Workload for rocprofv3 --pmc passes: the level-0 3x3 convolution (81920 x 320 x 2880) on the default routing (gemm16) and on
the halo-patch kernel (160-row and tall), three launches each -- separate kernels, so the counters come out per variant.
What to read (scripts/pmc_halo.sh, third pass): TCP_TCC_READ_REQ (L2 requests from the CUs), TCC_HIT / TCC_MISS -- the
prediction of docs/LAB_NOTEBOOK.md 5.5 (11) is that the halo kernel removes most of A's L2 misses, not just its requests."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tooncrafter_amd import ops
hip = ops.backend(); dev = "cuda"; BF = torch.bfloat16
def rnd(*s, scale=1.0): return (torch.randn(*s, device=dev) * scale).to(BF)
frames, h, w_, cin, cout = 32, 40, 64, 320, 320
x, wt, b = rnd(frames * h * w_, cin), rnd(cout, 9 * cin, scale=(9 * cin) ** -0.5), torch.randn(cout, device=dev)
geom = dict(kind="3x3", frames=frames, cin=cin, h_in=h, w_in=w_, h_out=h, w_out=w_, stride=1, upsample=False)
for halo, tall in (("0", "0"), ("2", "0"), ("2", "2")):
    os.environ["TC_CONV_HALO"], os.environ["TC_CONV_HALO_TALL"] = halo, tall
    for _ in range(3):
        hip.gemm(x, wt, b, conv=geom)
    torch.cuda.synchronize()
