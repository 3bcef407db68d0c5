#!/usr/bin/env python3
"""usage: lib_ab.py <other libtooncrafter_hip.so>  -- the in-tree library against another BUILD of it (e.g. the previous
commit's, kept under scripts/bin/prev/), interleaved in one process on the UNet's GEMM shapes at B = 2 under the default
routing; checks that both builds give the same bits."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tooncrafter_amd import _lib, ops
from tooncrafter_amd._lib import ACT_GEGLU, ACT_NONE
dev, BF = "cuda", torch.bfloat16
hip = ops.backend()
new = hip.lib
old = C.CDLL(os.path.abspath(sys.argv[1]))
for name, (res, args) in _lib.SYMBOLS.items():
    fn = getattr(old, name); fn.restype = res; fn.argtypes = args
assert old.tc_abi_version() == new.tc_abi_version()

def timeit(fn, iters=20, reps=3):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / iters)
    return min(ts)

def ab(fn, flops, tag):
    r, outs = {"old": [], "new": []}, {}
    for _ in range(2):
        for name, lib in (("old", old), ("new", new)):
            hip.lib = lib
            outs[name] = fn()
            r[name].append(timeit(fn))
    hip.lib = new
    same = torch.equal(outs["old"], outs["new"])
    a, b = min(r["old"]) * 1e3, min(r["new"]) * 1e3
    print(f"{tag:40s} old {a:7.1f} us {flops / a / 1e6:7.1f} TF/s | new {b:7.1f} us {flops / b / 1e6:7.1f} TF/s | x{a / b:5.3f} | {'same bits' if same else 'DIFFERENT BITS'}", flush=True)

def lin(m, n, k, tag, act=ACT_NONE, res=True):
    a = torch.randn(m, k, device=dev).to(BF); w = (torch.randn(n, k, device=dev) * k ** -0.5).to(BF)
    b = torch.randn(n, device=dev); r = torch.randn(m, n, device=dev).to(BF) if res and act != ACT_GEGLU else None
    ab(lambda: hip.gemm(a, w, b, act=act, residual=r), 2.0 * m * n * k, f"linear {tag} {m}x{n}x{k}")

def conv(frames, h, w, cin, cout, tag, t3=False):
    x = torch.randn(frames * h * w, cin, device=dev).to(BF)
    taps = 3 if t3 else 9
    wt = (torch.randn(cout, taps * cin, device=dev) * (taps * cin) ** -0.5).to(BF); b = torch.randn(cout, device=dev)
    res = torch.randn(frames * h * w, cout, device=dev).to(BF)
    geom = dict(kind="t3", frames=frames, t_len=16, cin=cin, h_out=h, w_out=w) if t3 else \
        dict(kind="3x3", frames=frames, cin=cin, h_in=h, w_in=w, h_out=h, w_out=w, stride=1, upsample=False)
    ab(lambda: hip.gemm(x, wt, b, conv=geom, residual=res), 2.0 * frames * h * w * cout * taps * cin,
       f"{'convT3' if t3 else 'conv3x3'} {tag} {cin}->{cout}")

lin(81920, 960, 320, "L0 qkv", res=False); lin(81920, 640, 320, "L0 640"); lin(20480, 640, 640, "L1 proj"); lin(20480, 1920, 640, "L1 qkv", res=False)
lin(20480, 640, 2560, "L1 ff2"); lin(5120, 1280, 1280, "L2 proj"); lin(5120, 3840, 1280, "L2 qkv", res=False); lin(5120, 1280, 5120, "L2 ff2")
lin(5120, 10240, 1280, "L2 geglu", act=ACT_GEGLU); lin(1280, 1280, 1280, "L3 proj"); lin(777, 520, 1288, "ragged")
conv(32, 10, 16, 1280, 1280, "L2"); conv(32, 10, 16, 2560, 1280, "L2"); conv(32, 5, 8, 1280, 1280, "L3")
conv(32, 10, 16, 1280, 1280, "L2", t3=True); conv(32, 5, 8, 1280, 1280, "L3", t3=True)
conv(16, 40, 64, 512, 512, "decoder")
