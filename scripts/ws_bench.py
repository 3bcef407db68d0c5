#!/usr/bin/env python3
"""A/B of the weight-stationary K = 320 kernel (csrc/gemm_ws.hip, TC_GEMM_WS) against the tiled kernels it replaces,
and the hipBLASLt YARDSTICK (torch.matmul / F.linear -- script only, never product) on the linear and square shapes of
profiles/r02_tile16_ab.txt.  One process, interleaved rounds (cdna_hip_programming.md 5.4 rule 24), random bf16 data,
HIP events on the launch stream, cache-cold rotation over several operand sets (a forward never re-reads the same
activations back to back).

    python scripts/ws_bench.py > gpurun_out/ws_bench.txt
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tooncrafter_amd import ops  # noqa: E402
from tooncrafter_amd._lib import ACT_GEGLU, ACT_NONE  # noqa: E402
from tooncrafter_amd.lvdm.common import pack_geglu  # noqa: E402

DEV, BF = "cuda", torch.bfloat16
hip = ops.backend()
SETS = 4          # operand sets rotated per launch: 4 x (52 + 52..210 MB) defeats the 256 MB Infinity Cache for L0


def time_variants(variants, iters=12, rounds=5):
    """variants: {name: fn(set_index)} -> {name: (median us, min us)}; rounds interleaved over the variants."""
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
    return {k: (sorted(v)[len(v) // 2], min(v)) for k, v in res.items()}


def env(ws):
    os.environ["TC_GEMM_WS"] = str(ws)


def linear_case(tag, m, n, k, geglu=False, res=False, ln=False, yard=True):
    a = [torch.randn(m, k, device=DEV).to(BF) for _ in range(SETS)]
    w32 = torch.randn(n, k, device=DEV) * k ** -0.5
    b32 = torch.randn(n, device=DEV)
    if geglu:
        w, b = pack_geglu(w32, b32)
    else:
        w, b = w32.to(BF), b32
    n_out = n // 2 if geglu else n
    r = [torch.randn(m, n_out, device=DEV).to(BF) for _ in range(SETS)] if res else [None] * SETS
    out = torch.empty(m, n_out, device=DEV, dtype=BF)
    act = ACT_GEGLU if geglu else ACT_NONE
    g = torch.ones(k, device=DEV)
    be = torch.zeros(k, device=DEV)
    flop = 2.0 * m * n * k

    def tiled(i):
        env(0)
        x = hip.layernorm(a[i], g, be, 1e-5) if ln else a[i]
        hip.gemm(x, w, b, act=act, residual=r[i], out=out)

    def ws(i, mode=2):
        env(mode)
        if ln:
            hip.gemm(a[i], w, b, act=act, residual=r[i], out=out, a_norm_eps=1e-5)
        else:
            hip.gemm(a[i], w, b, act=act, residual=r[i], out=out)

    variants = {"tiled": tiled}
    env(2)
    if k == 320 and hip.gemm_ln_eligible(m, n, k, geglu=geglu):
        variants["ws"] = ws
        variants["ws deep store window"] = lambda i: ws(i, 4)
    if yard:
        wb = w32.to(BF)

        def blaslt(i):
            y = torch.nn.functional.linear(a[i], wb)
            if geglu:
                v, gate = y.chunk(2, -1)
                y = v * torch.nn.functional.gelu(gate)
            if res:
                y = y + r[i]
            return y
        variants["hipBLASLt matmul only" if not (geglu or res or ln) else "torch eager (hipBLASLt + elementwise)"] = blaslt
        if geglu or res or ln:
            variants["hipBLASLt matmul only"] = lambda i: torch.nn.functional.linear(a[i], wb)
    t = time_variants(variants)
    env(1)
    cells = "  |  ".join(f"{name} {med:8.1f} us (min {mn:7.1f}) {flop / med / 1e6:7.1f} TF/s" for name, (med, mn) in t.items())
    extra = f"  x{t['tiled'][0] / t['ws'][0]:.2f} / x{t['tiled'][0] / t['ws deep store window'][0]:.2f}" if "ws" in t else ""
    print(f"{tag:34s} {m}x{n}x{k}{' +LN' if ln else ''}{' +res' if res else ''}{' GEGLU' if geglu else ''}: {cells}{extra}", flush=True)


if __name__ == "__main__":
    print(hip.lib.tc_build_info().decode(), torch.cuda.get_device_name(0))
    print("# level 0 (K = 320): weight-stationary kernel vs tiled kernel vs hipBLASLt")
    linear_case("L0 proj", 81920, 320, 320)
    linear_case("L0 proj + residual", 81920, 320, 320, res=True)
    linear_case("L0 LN + q", 81920, 320, 320, ln=True)
    linear_case("L0 qkv", 81920, 960, 320)
    linear_case("L0 LN + qkv", 81920, 960, 320, ln=True)
    linear_case("L0 GEGLU", 81920, 2560, 320, geglu=True)
    linear_case("L0 LN + GEGLU", 81920, 2560, 320, geglu=True, ln=True)
    linear_case("L0 proj B=1", 40960, 320, 320, res=True)
    print("# yardstick on the other linear shapes (tiled kernels vs hipBLASLt)")
    linear_case("L0 ff2", 81920, 320, 1280, res=True)
    linear_case("L1 proj", 20480, 640, 640, res=True)
    linear_case("L1 qkv", 20480, 1920, 640)
    linear_case("L1 GEGLU", 20480, 5120, 640, geglu=True)
    linear_case("L1 ff2", 20480, 640, 2560, res=True)
    linear_case("L2 proj", 5120, 1280, 1280, res=True)
    linear_case("L2 qkv", 5120, 3840, 1280)
    linear_case("L2 GEGLU", 5120, 10240, 1280, geglu=True)
    linear_case("L2 ff2", 5120, 1280, 5120, res=True)
    linear_case("L3 proj", 1280, 1280, 1280, res=True)
    linear_case("L3 qkv", 1280, 3840, 1280)
    linear_case("square 4k (N = 4000)", 4096, 4000, 4096)
    linear_case("square 4096", 4096, 4096, 4096)
    linear_case("square 8192", 8192, 8192, 8192)
