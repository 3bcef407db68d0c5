#!/usr/bin/env python3
"""Time the one-launch level-0 temporal self-attention (csrc/tb_fused.hip) against the launches it replaces.

    python scripts/tb_fused_bench.py            # B = 2, 16 frames, 40 x 64 pixels (BASELINE level 0)
"""
import os
import sys

import torch

os.environ.setdefault("TC_TB_FUSED", "1")   # opt-in since round 6 (the level-0 default is the chain around csrc/qkv_attn.hip)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tooncrafter_amd.lvdm.common import pack_linear  # noqa: E402
from tooncrafter_amd.ops import HipOps  # noqa: E402

C, HEADS, T = 320, 5, 16


def timeit(fn, n=20, warm=3):
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
    b, hw = 2, 2560
    wqkv = pack_linear(torch.randn(3 * C, C, generator=g) * 0.06).cuda()
    bqkv = (torch.randn(3 * C, generator=g) * 0.1).cuda()
    wo, bo = pack_linear(torch.randn(C, C, generator=g) * 0.05).cuda(), (torch.randn(C, generator=g) * 0.1).cuda()
    ones, zeros = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    x = (torch.randn(b * T * hw, C, generator=g) * 1.5).to(torch.bfloat16).cuda()

    def chain():
        qkv = hip.gemm(hip.layernorm(x, ones, zeros, 1e-5), wqkv, bqkv)
        a = hip.attention_temporal(qkv, b=b, t=T, hw=hw, heads=HEADS)
        return hip.gemm(a, wo, bo, residual=x)

    fused = lambda: hip.temporal_attn_fused(x, wqkv, bqkv, wo, bo, b=b, t=T, hw=hw, heads=HEADS, ln_eps=1e-5)
    h = hip.layernorm(x, ones, zeros, 1e-5)
    qkv = hip.gemm(h, wqkv, bqkv)
    att = hip.attention_temporal(qkv, b=b, t=T, hw=hw, heads=HEADS)
    print(f"LayerNorm {timeit(lambda: hip.layernorm(x, ones, zeros, 1e-5)):.1f} us | qkv {timeit(lambda: hip.gemm(h, wqkv, bqkv)):.1f} us | "
          f"attention {timeit(lambda: hip.attention_temporal(qkv, b=b, t=T, hw=hw, heads=HEADS)):.1f} us | "
          f"out-proj + residual {timeit(lambda: hip.gemm(att, wo, bo, residual=x)):.1f} us")
    arms = [("chain", {}, chain), ("fused (first after the chain)", {}, fused), ("fused", {}, fused)]
    for var, key in (("TC_TB_GRID", "TB_GRIDS"), ("TC_TB_STAGGER", "TB_STAGGERS"), ("TC_TB_ABLATE", "TB_ABLS")):
        for v in [v for v in os.environ.get(key, "").split(",") if v]:
            arms.append((f"fused {var}={v}", {var: v}, fused))
    arms += [("fused (again)", {}, fused), ("chain (again, first)", {}, chain), ("chain (again)", {}, chain)]
    times = [[] for _ in arms]
    for rd in range(4):
        for i, (_, env, fn) in enumerate(arms):
            os.environ.update(env)
            t = timeit(fn)
            for k in env:
                os.environ.pop(k)
            if rd:
                times[i].append(t)
    med = [sorted(t)[len(t) // 2] for t in times]
    d = (fused().float() - chain().float()).abs().max()
    for (name, _, _), t in zip(arms, med):
        print(f"{name:34s} {t:7.1f} us  x{med[0] / t:.3f} vs chain")
    print(f"max |fused - chain| {float(d):.3e}")


if __name__ == "__main__":
    main()
