#!/usr/bin/env python3
"""A/B of the K loop of csrc/gemm.hip: TC_GEMM_PIPE=0 (one K-step of LDS-DMA in flight, vmcnt(0) + __syncthreads per
step) against TC_GEMM_PIPE=1 (two in flight, counted vmcnt + raw barriers, epilogue operands prefetched), on the
shapes of a B = 2 UNet forward that the 128x128 / 64x64 kernels serve.  One process, interleaved rounds, operands
rotated over sets that together exceed the Infinity Cache.

    python scripts/pipe_bench.py > gpurun_out/pipe_bench.txt
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
SETS = 4


def time_variants(variants, iters=16, rounds=5):
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


def case(tag, m, n, k, geglu=False, res=False, conv=None, a_rows=None, extra_env=None):
    a_rows = a_rows or m
    kk = k * {"3x3": 9, "t3": 3}.get(conv["kind"], 1) if conv else k
    a = [torch.randn(a_rows, k, device=DEV).to(BF) for _ in range(SETS)]
    w32 = torch.randn(n, kk, device=DEV) * kk ** -0.5
    b32 = torch.randn(n, device=DEV)
    w, b = pack_geglu(w32, b32) if geglu else (w32.to(BF), b32)
    n_out = n // 2 if geglu else n
    r = [torch.randn(m, n_out, device=DEV).to(BF) for _ in range(SETS)] if res else [None] * SETS
    out = torch.empty(m, n_out, device=DEV, dtype=BF)
    act = ACT_GEGLU if geglu else ACT_NONE

    def run(i, pipe, late="0"):
        os.environ["TC_GEMM_PIPE"] = pipe
        os.environ["TC_GEMM_EPI_LATE"] = late
        hip.gemm(a[i], w, b, act=act, residual=r[i], out=out, conv=conv)

    os.environ["TC_GEMM_WS"] = "0"
    for key, val in (extra_env or {}).items():
        os.environ[key] = val
    t = time_variants({"r02": lambda i: run(i, "0", "1"), "plain": lambda i: run(i, "0"), "pipelined": lambda i: run(i, "2")})
    for key in (extra_env or {}):
        os.environ.pop(key)
    fl = 2.0 * m * n * kk
    print(f"{tag:30s} {m}x{n}x{kk}{' +res' if res else ''}{' GEGLU' if geglu else ''}: r02 loop, late epilogue loads {t['r02']:8.1f} us | early epilogue loads {t['plain']:8.1f} us "
          f"{fl / t['plain'] / 1e6:7.1f} TF/s | pipelined {t['pipelined']:8.1f} us {fl / t['pipelined'] / 1e6:7.1f} TF/s | "
          f"x{t['plain'] / t['pipelined']:.3f}", flush=True)


def c3(frames, h, w, cin):
    return dict(kind="3x3", frames=frames, cin=cin, h_in=h, w_in=w, h_out=h, w_out=w, stride=1, upsample=False)


def ct(frames, hw, cin):
    return dict(kind="t3", frames=frames, t_len=16, cin=cin, h_out=1, w_out=hw)


if __name__ == "__main__":
    print(hip.lib.tc_build_info().decode(), torch.cuda.get_device_name(0))
    case("L0 proj + res", 81920, 320, 320, res=True)
    case("L0 qkv", 81920, 960, 320)
    case("L0 GEGLU", 81920, 2560, 320, geglu=True)
    case("L0 ff2", 81920, 320, 1280, res=True)
    case("L1 proj + res", 20480, 640, 640, res=True)
    case("L1 qkv", 20480, 1920, 640)
    case("L1 GEGLU", 20480, 5120, 640, geglu=True)
    case("L1 ff2", 20480, 640, 2560, res=True)
    case("L2 proj + res", 5120, 1280, 1280, res=True)
    case("L2 qkv", 5120, 3840, 1280)
    case("L2 ff2", 5120, 1280, 5120, res=True)
    case("L3 proj + res (64x64 tiles)", 1280, 1280, 1280, res=True)
    case("L3 qkv", 1280, 3840, 1280)
    no16 = {"TC_GEMM_TILE16": "0"}
    case("conv3x3 L0 320->320 (128-tile)", 81920, 320, 320, conv=c3(32, 40, 64, 320), extra_env=no16)
    case("conv3x3 L1 640->640 (128-tile)", 20480, 640, 640, conv=c3(32, 20, 32, 640), extra_env=no16)
    case("conv3x3 L2 1280->1280", 5120, 1280, 1280, conv=c3(32, 10, 16, 1280), res=True)
    case("conv3x3 L2 2560->1280", 5120, 1280, 2560, conv=c3(32, 10, 16, 2560))
    case("convT3 L0 320->320 (128-tile)", 81920, 320, 320, conv=ct(32, 2560, 320), res=True, extra_env=no16)
    case("convT3 L2 1280->1280", 5120, 1280, 1280, conv=ct(32, 160, 1280), res=True)
    case("convT3 L3 1280->1280", 1280, 1280, 1280, conv=ct(32, 40, 1280), res=True)
    case("square 4096", 4096, 4096, 4096)
    print("# 160x160 kernel (gemm16.hip) and 256-row kernel (gemm_wide.hip)")
    case("conv3x3 L0 320->320 (tile16)", 81920, 320, 320, conv=c3(32, 40, 64, 320), res=True)
    case("conv3x3 L0 960->320 (tile16)", 81920, 320, 960, conv=c3(32, 40, 64, 960))
    case("conv3x3 L1 640->640 (tile16)", 20480, 640, 640, conv=c3(32, 20, 32, 640), res=True)
    case("conv3x3 L1 1280->640 (tile16)", 20480, 640, 1280, conv=c3(32, 20, 32, 1280))
    case("convT3 L0 320->320 (tile16)", 81920, 320, 320, conv=ct(32, 2560, 320), res=True)
    case("convT3 L1 640->640 (tile16)", 20480, 640, 640, conv=ct(32, 640, 640), res=True)
    case("L2 GEGLU (wide)", 5120, 10240, 1280, geglu=True)
    case("decoder conv3x3 128->128 4f (wide)", 4 * 320 * 512, 128, 128, conv=c3(4, 320, 512, 128))
    case("decoder conv3x3 256->256 4f (wide)", 4 * 160 * 256, 256, 256, conv=c3(4, 160, 256, 256))
    case("decoder conv3x3 512->512 16f (wide)", 16 * 80 * 128, 512, 512, conv=c3(16, 80, 128, 512))
