#!/usr/bin/env python3
"""A/B of the four-wave 256x256 kernel (csrc/gemm4.hip, TC_GEMM4=2) against the default routing (TC_GEMM4=0) and the hipBLASLt
yardstick (torch F.linear -- script only), one process, interleaved rounds, operands rotated over 4 sets (cache-cold)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from tooncrafter_amd import ops  # noqa: E402
from tooncrafter_amd._lib import ACT_GEGLU, ACT_NONE  # noqa: E402
from tooncrafter_amd.lvdm.common import pack_geglu  # noqa: E402

DEV, BF, SETS = "cuda", torch.bfloat16, 4
from tooncrafter_amd.ops import HipOps  # noqa: E402
hip = HipOps()


def time_variants(variants, iters=10, rounds=5):
    for fn in variants.values():
        fn(0)
    torch.cuda.synchronize()
    res = {k: [] for k in variants}
    for _ in range(rounds):
        for name, fn in variants.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(iters):
                fn(i % SETS)
            e1.record()
            torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) / iters * 1e3)
    return {k: sorted(v)[len(v) // 2] for k, v in res.items()}


def case(tag, m, n, k, geglu=False, res=False):
    a = [torch.randn(m, k, device=DEV).to(BF) for _ in range(SETS)]
    if geglu:
        wp, bp = pack_geglu(torch.randn(n, k) * k ** -0.5, torch.randn(n) * 0.1)
        w, b = wp.to(DEV), bp.to(DEV)
    else:
        w, b = (torch.randn(n, k, device=DEV) * k ** -0.5).to(BF), torch.randn(n, device=DEV)
    r = [torch.randn(m, n, device=DEV).to(BF) for _ in range(SETS)] if res else None
    act = ACT_GEGLU if geglu else ACT_NONE

    def mine(mode, variant="2"):
        def f(i):
            os.environ["TC_GEMM4"], os.environ["TC_G4_VARIANT"] = mode, variant
            return hip.gemm(a[i], w, b, act=act, residual=None if r is None else r[i])
        return f
    v = {"default routing": mine("0"), "gemm4 v0 (burst reads)": mine("2", "0"), "gemm4 v1 (1 barrier)": mine("2", "1"),
         "gemm4": mine("2", "2"), "hipBLASLt matmul only": lambda i: F.linear(a[i], w)}
    t = time_variants(v)
    os.environ.pop("TC_GEMM4", None)
    fl = 2.0 * m * n * k
    print(f"{tag:28s} {m}x{n}x{k}{' GEGLU' if geglu else ''}{' +res' if res else ''}: " +
          " | ".join(f"{nm} {us:8.1f} us {fl / us / 1e6:7.1f} TF/s" for nm, us in t.items()) +
          f" | gemm4 / default x{t['default routing'] / t['gemm4']:.3f}", flush=True)


case("square 8192", 8192, 8192, 8192)
case("square 4096", 4096, 4096, 4096)
case("L1 GEGLU", 20480, 5120, 640, geglu=True)
case("L2 GEGLU", 5120, 10240, 1280, geglu=True)
case("L1 ff2", 20480, 640, 2560, res=True)
case("L2 ff2", 5120, 1280, 5120, res=True)
case("L1 qkv", 20480, 1920, 640)
case("L2 qkv", 5120, 3840, 1280)
case("L0 qkv", 81920, 960, 320)
case("L0 GEGLU", 81920, 2560, 320, geglu=True)
case("L2 proj", 5120, 1280, 1280, res=True)
case("dec mid attn qkv-like", 40960, 1536, 512)
