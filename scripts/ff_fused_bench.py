#!/usr/bin/env python3
"""Time the one-launch level-0 feed-forward (csrc/ff_fused.hip) against the launches it replaces, on the GPU box.

    python scripts/ff_fused_bench.py [rows ...]        # default 81920 (BASELINE level 0, both guidance passes)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tooncrafter_amd._lib import ACT_GEGLU  # noqa: E402
from tooncrafter_amd.lvdm.common import pack_geglu, pack_linear  # noqa: E402
from tooncrafter_amd.ops import HipOps  # noqa: E402

C, HID = 320, 1280


def timeit(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    hip = HipOps()
    g = torch.Generator().manual_seed(0)
    w1, b1 = pack_geglu(torch.randn(2 * HID, C, generator=g) * 0.05, torch.randn(2 * HID, generator=g) * 0.1)
    w2, b2 = pack_linear(torch.randn(C, HID, generator=g) * 0.03), torch.randn(C, generator=g) * 0.1
    w1, b1, w2, b2 = w1.cuda(), b1.cuda(), w2.cuda(), b2.cuda()
    ones, zeros = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    for m in [int(a) for a in sys.argv[1:]] or [81920]:
        x = (torch.randn(m, C, generator=g) * 1.5).to(torch.bfloat16).cuda()
        flops = 2.0 * m * C * HID * 3

        def chain():
            h = hip.layernorm(x, ones, zeros, 1e-5)
            gg = hip.gemm(h, w1, b1, act=ACT_GEGLU)
            return hip.gemm(gg, w2, b2, residual=x)

        t_ln = timeit(lambda: hip.layernorm(x, ones, zeros, 1e-5))
        h = hip.layernorm(x, ones, zeros, 1e-5)
        t_g = timeit(lambda: hip.gemm(h, w1, b1, act=ACT_GEGLU))
        gg = hip.gemm(h, w1, b1, act=ACT_GEGLU)
        t_2 = timeit(lambda: hip.gemm(gg, w2, b2, residual=x))
        t_chain = timeit(chain)
        print(f"rows {m}: LayerNorm {t_ln:.1f} us | GEGLU projection {t_g:.1f} us | ff2 + residual {t_2:.1f} us | chain {t_chain:.1f} us")
        for grid in [int(v) for v in os.environ.get("FF_GRIDS", "0").split(",")]:
            if grid:
                os.environ["TC_FF_GRID"] = str(grid)
            else:
                os.environ.pop("TC_FF_GRID", None)
            t_f = timeit(lambda: hip.ff_geglu_fused(x, w1, b1, w2, b2, ln_eps=1e-5))
            for var, vals in (("TC_FF_LOOKAHEAD", os.environ.get("FF_LAS", "")), ("TC_FF_ABLATE", os.environ.get("FF_ABLS", ""))):
                for v in [v for v in vals.split(",") if v]:
                    os.environ[var] = v
                    t_v = timeit(lambda: hip.ff_geglu_fused(x, w1, b1, w2, b2, ln_eps=1e-5))
                    print(f"rows {m}:   {var}={v}: {t_v:.1f} us")
                os.environ.pop(var, None)
            d = (hip.ff_geglu_fused(x, w1, b1, w2, b2, ln_eps=1e-5).float() - chain().float()).abs().max()
            print(f"rows {m}: fused (grid {grid or 'CUs'}) {t_f:.1f} us = {flops / t_f * 1e-6:.1f} TFLOP/s, "
                  f"{t_chain / t_f:.2f}x the chain; max |fused - chain| {float(d):.3e}")


if __name__ == "__main__":
    main()
