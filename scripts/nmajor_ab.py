"""A/B of the W-stationary tile walk (TC_GEMM_NMAJOR = 0 never / 1 heuristic / 2 always, read per call) on the
low-resolution layers of one B=2 UNet forward: same process, interleaved; results must be bit-identical."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tooncrafter_amd import ops  # noqa: E402
from tooncrafter_amd._lib import ACT_GEGLU, ACT_NONE  # noqa: E402

hip = ops.backend()
BF = torch.bfloat16
MODES = [0, 1, 2]


def timed(fn, iters=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def ab(tag, fn, flops):
    outs, ts = {}, {m: [] for m in MODES}
    for m in MODES:
        os.environ["TC_GEMM_NMAJOR"] = str(m)
        outs[m] = fn().clone()
    same = all(torch.equal(outs[0], outs[m]) for m in MODES)
    for _ in range(5):
        for m in MODES:
            os.environ["TC_GEMM_NMAJOR"] = str(m)
            ts[m].append(timed(fn))
    med = {m: sorted(v)[2] for m, v in ts.items()}
    print(f"{tag:28s} " + " ".join(f"{med[m]:8.1f}" for m in MODES) + f"   {flops / med[0] * 1e-6:6.0f} -> {flops / med[1] * 1e-6:6.0f} TF/s (heuristic)"
          f"  x{med[0] / med[1]:.2f}  identical={same}", flush=True)


def lin(m, n, k, act=ACT_NONE, res=False, tag=""):
    a = torch.randn(m, k, device="cuda").to(BF)
    w = (torch.randn(n, k, device="cuda") * k ** -0.5).to(BF)
    b = torch.randn(n, device="cuda")
    r = torch.randn(m, n, device="cuda").to(BF) if res else None
    ab(tag, lambda: hip.gemm(a, w, b, act=act, residual=r), 2.0 * m * n * k)


def conv(frames, h, w, cin, cout, tag="", t3=False):
    x = torch.randn(frames * h * w, cin, device="cuda").to(BF)
    taps = 3 if t3 else 9
    wt = (torch.randn(cout, taps * cin, device="cuda") * (taps * cin) ** -0.5).to(BF)
    b = torch.randn(cout, device="cuda")
    geom = dict(kind="t3", frames=frames, t_len=16, cin=cin, h_out=h, w_out=w) if t3 else \
        dict(kind="3x3", frames=frames, cin=cin, h_in=h, w_in=w, h_out=h, w_out=w, stride=1, upsample=False)
    ab(tag, lambda: hip.gemm(x, wt, b, conv=geom), 2.0 * frames * h * w * cout * taps * cin)


print(f"{'shape':28s} " + " ".join(f"{'mode ' + str(m):>8s}" for m in MODES) + "   (us, median of 5 interleaved rounds)")
conv(32, 5, 8, 1280, 1280, "L3 conv 1280->1280")
conv(32, 5, 8, 2560, 1280, "L3 conv 2560->1280")
conv(32, 5, 8, 1280, 1280, "L3 tconv", t3=True)
lin(1280, 1280, 1280, res=True, tag="L3 proj")
lin(1280, 3840, 1280, tag="L3 qkv")
lin(1280, 10240, 1280, act=ACT_GEGLU, tag="L3 geglu")
lin(1280, 1280, 5120, res=True, tag="L3 ff2")
conv(32, 10, 16, 1280, 1280, "L2 conv 1280->1280")
conv(32, 10, 16, 2560, 1280, "L2 conv 2560->1280")
conv(32, 10, 16, 1920, 1280, "L2 conv 1920->1280")
conv(32, 10, 16, 640, 1280, "L2 conv 640->1280")
conv(32, 10, 16, 1280, 1280, "L2 tconv", t3=True)
lin(5120, 1280, 1280, res=True, tag="L2 proj")
lin(5120, 3840, 1280, tag="L2 qkv")
lin(5120, 10240, 1280, act=ACT_GEGLU, tag="L2 geglu")
lin(5120, 1280, 5120, res=True, tag="L2 ff2")
conv(32, 20, 32, 640, 640, "L1 conv 640->640")
lin(20480, 5120, 640, act=ACT_GEGLU, tag="L1 geglu")
lin(2, 1280, 320, tag="time embed")
lin(666, 1280, 1024, tag="context k/v")
