#!/usr/bin/env python3
"""Microbenchmark of tc_gemm_bf16 on the shapes that dominate one UNet forward (B=2) and the
decoder.  Random bf16 data (zero-filled operands clock ~20 % faster on this chip), HIP events
on the launch stream, median of several batches.  Prints TFLOP/s per shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tooncrafter_amd import ops
from tooncrafter_amd._lib import ACT_GEGLU, ACT_NONE

dev = "cuda"
hip = ops.backend()
BF = torch.bfloat16


def timeit(fn, iters=20, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters)
    return sorted(ts)[len(ts) // 2]


def lin(m, n, k, act=ACT_NONE, res=False, tag=""):
    a = torch.randn(m, k, device=dev).to(BF)
    w = (torch.randn(n, k, device=dev) * k ** -0.5).to(BF)
    b = torch.randn(n, device=dev)
    r = torch.randn(m, n, device=dev).to(BF) if res else None
    ms = timeit(lambda: hip.gemm(a, w, b, act=act, residual=r))
    fl = 2.0 * m * n * k
    print(f"linear  {tag:14s} M={m:6d} N={n:5d} K={k:5d} act={act} res={int(res)}  {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s")


def conv(frames, h, w, cin, cout, tag="", t3=False, tlen=16):
    x = torch.randn(frames * h * w, cin, device=dev).to(BF)
    taps = 3 if t3 else 9
    wt = (torch.randn(cout, taps * cin, device=dev) * (taps * cin) ** -0.5).to(BF)
    b = torch.randn(cout, device=dev)
    geom = dict(kind="t3", frames=frames, t_len=tlen, cin=cin, h_out=h, w_out=w) if t3 else \
        dict(kind="3x3", frames=frames, cin=cin, h_in=h, w_in=w, h_out=h, w_out=w, stride=1, upsample=False)
    ms = timeit(lambda: hip.gemm(x, wt, b, conv=geom))
    fl = 2.0 * frames * h * w * cout * taps * cin
    print(f"{'convT3 ' if t3 else 'conv3x3'} {tag:14s} f={frames:3d} {h}x{w} {cin}->{cout}  {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s")


if __name__ == "__main__" and "--attn-only" not in sys.argv:
    print(hip.lib.tc_build_info().decode())
    lin(81920, 320, 320, res=True, tag="L0 proj")
    lin(81920, 960, 320, tag="L0 qkv")
    lin(81920, 2560, 320, act=ACT_GEGLU, tag="L0 geglu")
    lin(81920, 320, 1280, res=True, tag="L0 ff2")
    lin(20480, 640, 640, res=True, tag="L1 proj")
    lin(20480, 5120, 640, act=ACT_GEGLU, tag="L1 geglu")
    lin(20480, 640, 2560, res=True, tag="L1 ff2")
    lin(5120, 1280, 1280, res=True, tag="L2 proj")
    lin(5120, 10240, 1280, act=ACT_GEGLU, tag="L2 geglu")
    lin(5120, 1280, 5120, res=True, tag="L2 ff2")
    lin(8192, 8192, 8192, tag="square 8k")
    lin(4096, 4096, 4096, tag="square 4k")
    conv(32, 40, 64, 320, 320, "L0 res")
    conv(32, 20, 32, 640, 640, "L1 res")
    conv(32, 10, 16, 1280, 1280, "L2 res")
    conv(32, 5, 8, 1280, 1280, "L3 res")
    conv(32, 40, 64, 320, 320, "L0 tconv", t3=True)
    conv(32, 10, 16, 1280, 1280, "L2 tconv", t3=True)
    conv(16, 80, 128, 512, 512, "dec L2")
    conv(16, 320, 512, 128, 128, "dec L0")
    if "--more" in sys.argv:     # further shapes of the UNet (B=2) and decoder for the tile sweep
        lin(20480, 1920, 640, tag="L1 qkv")
        lin(5120, 3840, 1280, tag="L2 qkv")
        lin(1280, 1280, 1280, res=True, tag="L3 proj")
        lin(1280, 3840, 1280, tag="L3 qkv")
        lin(1280, 10240, 1280, act=ACT_GEGLU, tag="L3 geglu")
        lin(1280, 1280, 5120, res=True, tag="L3 ff2")
        conv(32, 40, 64, 640, 320, "L0 res 640in")
        conv(32, 40, 64, 960, 320, "L0 res 960in")
        conv(32, 20, 32, 1280, 640, "L1 res 1280in")
        conv(32, 20, 32, 320, 640, "L1 res 320in")
        conv(32, 10, 16, 2560, 1280, "L2 res 2560in")
        conv(32, 5, 8, 2560, 1280, "L3 res 2560in")
        conv(32, 20, 32, 640, 640, "L1 tconv", t3=True)
        conv(32, 5, 8, 1280, 1280, "L3 tconv", t3=True)
        conv(16, 40, 64, 512, 512, "dec L3")
        conv(16, 160, 256, 256, 256, "dec L1")
        conv(16, 160, 256, 512, 256, "dec L1 512in")
        conv(16, 320, 512, 256, 128, "dec L0 256in")
    if "--no-attn" in sys.argv:
        sys.exit(0)


def attn(batch, heads, lq, lk, kv_bdiv=1, tag=""):
    c = heads * 64
    q = torch.randn(batch * lq, c, device=dev).to(BF)
    kvb = (batch + kv_bdiv - 1) // kv_bdiv
    kv = torch.randn(kvb * lk, 2 * c, device=dev).to(BF)
    ms = timeit(lambda: hip.attention(q, kv[:, :c], kv[:, c:], batch=batch, heads=heads, lq=lq, lk=lk, kv_bdiv=kv_bdiv))
    fl = 4.0 * batch * heads * lq * lk * 64
    print(f"attn    {tag:14s} b={batch} h={heads} lq={lq} lk={lk}  {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s")


if __name__ == "__main__":
    attn(32, 5, 2560, 2560, tag="L0 self")
    attn(32, 10, 640, 640, tag="L1 self")
    attn(32, 20, 160, 160, tag="L2 self")
    attn(32, 5, 2560, 77, 16, tag="L0 text")
    attn(32, 5, 2560, 16, 1, tag="L0 image")
    attn(16, 8, 10240, 20480, 16, tag="dec L2 ref")
    # text + image cross-attention as ONE launch (two softmaxes)
    c = 5 * 64
    q = torch.randn(32 * 2560, c, device=dev).to(BF)
    kvt, kvi = torch.randn(2 * 77, 2 * c, device=dev).to(BF), torch.randn(32 * 16, 2 * c, device=dev).to(BF)
    ms = timeit(lambda: hip.attention(q, kvt[:, :c], kvt[:, c:], batch=32, heads=5, lq=2560, lk=77, kv_bdiv=16,
                                      k2=kvi[:, :c], v2=kvi[:, c:], lk2=16, kv2_bdiv=1))
    print(f"attn    L0 text+image fused  b=32 h=5 lq=2560 lk=77+16  {ms*1e3:8.1f} us")
