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

        # arms, timed interleaved over several rounds (the clocks drift over the first seconds of load: an arm measured first
        # reads ~8 % slower than the same arm measured last) -- medians
        arms = [("chain: LayerNorm + GEGLU projection + ff2", {}, chain)]
        fused = lambda: hip.ff_geglu_fused(x, w1, b1, w2, b2, ln_eps=1e-5)
        arms.append(("fused (first after the chain: clocks in transit)", {}, fused))
        arms.append(("fused", {}, fused))
        for var, key in (("TC_FF_GRID", "FF_GRIDS"), ("TC_FF_LOOKAHEAD", "FF_LAS"), ("TC_FF_GILP", "FF_GILPS"), ("TC_FF_ABLATE", "FF_ABLS")):
            for v in [v for v in os.environ.get(key, "").split(",") if v]:
                arms.append((f"fused {var}={v}", {var: v}, fused))
        arms.append(("fused (again)", {}, fused))
        arms.append(("chain (again, first after fused)", {}, chain))
        arms.append(("chain (again)", {}, chain))
        times = [[] for _ in arms]
        for rd in range(int(os.environ.get("FF_ROUNDS", "5")) + 1):
            for i, (_, env, fn) in enumerate(arms):
                os.environ.update(env)
                t = timeit(fn, n=20, warm=3)
                for k in env:
                    os.environ.pop(k)
                if rd:
                    times[i].append(t)
        med = [sorted(t)[len(t) // 2] for t in times]
        d = (fused().float() - chain().float()).abs().max()
        for (name, _, _), t in zip(arms, med):
            print(f"rows {m}: {name:46s} {t:7.1f} us  {flops / t * 1e-6:6.1f} TFLOP/s  x{med[0] / t:.3f} vs chain")
        print(f"rows {m}: max |fused - chain| {float(d):.3e}")
        # ---- in context: what a level-0 block runs around its feed-forward (the attention's output projection + residual
        # before it, the next block's qkv projection after it -- both HBM-bound), events around the feed-forward only
        wo = pack_linear(torch.randn(C, C, generator=g) * 0.05).cuda()
        wqkv = pack_linear(torch.randn(3 * C, C, generator=g) * 0.05).cuda()
        att = (torch.randn(m, C, generator=g)).to(torch.bfloat16).cuda()

        def in_context(ff, reps=20):
            tot = 0.0
            for it in range(reps + 3):
                xx = hip.gemm(att, wo, None, residual=x)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                y = ff(xx)
                e1.record()
                hip.gemm(y, wqkv, None)
                torch.cuda.synchronize()
                if it >= 3:
                    tot += e0.elapsed_time(e1)
            return tot / reps * 1e3

        def ff_chain(xx):
            h = hip.layernorm(xx, ones, zeros, 1e-5)
            return hip.gemm(hip.gemm(h, w1, b1, act=ACT_GEGLU), w2, b2, residual=xx)

        for rd in range(3):
            print(f"rows {m}: in context (out-proj -> FF -> qkv), round {rd}: chain {in_context(ff_chain):7.1f} us | "
                  f"fused {in_context(lambda xx: hip.ff_geglu_fused(xx, w1, b1, w2, b2, ln_eps=1e-5)):7.1f} us")


if __name__ == "__main__":
    main()
