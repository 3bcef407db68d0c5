#!/usr/bin/env python3
"""GroupNorm + SiLU at the shapes the 768-thread one-pass instance (round 6, csrc/norm.hip) takes, TC_GN_ONEPASS_BIG = 0 / 1
interleaved in one process: alone on rotating tensors, and right behind a kernel that has just WRITTEN x (the situation
inside the forward: x comes out of the producing convolution and sits in the memory-side cache).
    python scripts/gn_big_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tooncrafter_amd.ops import HipOps  # noqa: E402


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
    print(f"# {hip.lib.tc_build_info().decode()}")
    for tag, samples, rows, c in (("level 0 per-frame", 32, 2560, 320), ("level 2 clip-wide", 2, 2560, 1280), ("level 0 rows, 640 ch", 32, 2560, 640),
                                  ("level 1 per-frame (256-thread one-pass either way)", 32, 640, 640), ("level 0 clip-wide (three launches either way)", 2, 40960, 320)):
        nrot = 8
        xs = [torch.randn(samples * rows, c, device="cuda").to(torch.bfloat16) for _ in range(nrot)]
        src = torch.randn(samples * rows, c, device="cuda").to(torch.bfloat16)
        g, b = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
        norm = lambda i: hip.groupnorm(xs[i % nrot], g, b, samples=samples, rows=rows, eps=1e-5, silu=True)
        def fresh(i):
            xs[i % nrot].copy_(src)                                  # a producer writes x ...
            return hip.groupnorm(xs[i % nrot], g, b, samples=samples, rows=rows, eps=1e-5, silu=True)     # ... the norm follows
        tcopy = timeit(lambda i: xs[i % nrot].copy_(src))
        res = {}
        for rd in range(3):
            for v in ("0", "1"):
                os.environ["TC_GN_ONEPASS_BIG"] = v
                res.setdefault(v, []).append((timeit(norm), timeit(fresh) - tcopy))
        os.environ.pop("TC_GN_ONEPASS_BIG")
        m = {v: (sorted(t[0] for t in r)[1], sorted(t[1] for t in r)[1]) for v, r in res.items()}
        os.environ["TC_GN_ONEPASS_BIG"] = "0"
        y0 = norm(0)
        os.environ["TC_GN_ONEPASS_BIG"] = "1"
        y1 = norm(0)
        os.environ.pop("TC_GN_ONEPASS_BIG")
        nb = 4.0 * samples * rows * c
        print(f"{tag:50s} s{samples} r{rows} c{c}: rotating {m['0'][0]:6.1f} -> {m['1'][0]:6.1f} us (x{m['0'][0] / m['1'][0]:.2f}, {nb / m['1'][0] * 1e-6:.2f} TB/s algorithmic) | "
              f"behind its producer {m['0'][1]:6.1f} -> {m['1'][1]:6.1f} us (x{m['0'][1] / m['1'][1]:.2f}) | max |d| {float((y0.float() - y1.float()).abs().max()):.2e}")


if __name__ == "__main__":
    main()
