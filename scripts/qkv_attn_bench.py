#!/usr/bin/env python3
"""Time the qkv projection + temporal attention launch (csrc/qkv_attn.hip, ABI 13) against the two launches it replaces, at
the BASELINE geometries of UNet levels 1 / 2 / 3 (B = 2, 16 frames), interleaved in one process, operands rotated through
more than the Infinity Cache so that neither arm runs on warm weights only.

    python scripts/qkv_attn_bench.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tooncrafter_amd.lvdm.common import pack_linear  # noqa: E402
from tooncrafter_amd.ops import HipOps  # noqa: E402

T = 16


def timeit(fn, n=20, warm=3):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    hip = HipOps()
    g = torch.Generator().manual_seed(0)
    print(f"# {hip.lib.tc_build_info().decode()}")
    for tag, b, hw, c in (("level 1", 2, 640, 640), ("level 2", 2, 160, 1280), ("level 3", 2, 40, 1280), ("level 0 width", 2, 2560, 320)):
        heads = c // 64
        nrot = 6
        ws = [pack_linear(torch.randn(3 * c, c, generator=g) * 1.4 * c ** -0.5).cuda() for _ in range(nrot)]
        xs = [(torch.randn(b * T * hw, c, generator=g) * 1.2).to(torch.bfloat16).cuda() for _ in range(nrot)]
        kw = dict(b=b, t=T, hw=hw, heads=heads)
        qkv0 = hip.gemm(xs[0], ws[0])
        two = lambda i: hip.attention_temporal(hip.gemm(xs[i % nrot], ws[i % nrot]), **kw)
        one = lambda i: hip.temporal_qkv_attn(xs[i % nrot], ws[i % nrot], None, **kw)
        tg = timeit(lambda i: hip.gemm(xs[i % nrot], ws[i % nrot]))
        ta = timeit(lambda i: hip.attention_temporal(qkv0, **kw))
        rounds = [(timeit(two), timeit(one)) for _ in range(4)][1:]
        t2 = sorted(r[0] for r in rounds)[1]
        t1 = sorted(r[1] for r in rounds)[1]
        fl = 2.0 * b * T * hw * 3 * c * c
        d = (one(0).float() - two(0).float()).abs().max()
        print(f"{tag:14s} rows {b * T * hw:6d} C {c:5d}: qkv GEMM {tg:6.1f} us + attention {ta:5.1f} us | two launches {t2:6.1f} us | "
              f"one launch {t1:6.1f} us ({fl / t1 * 1e-6:5.0f} TF/s on the projection's FLOPs)  x{t2 / t1:.3f}   max |d| {float(d):.2e}")


if __name__ == "__main__":
    main()
