#!/usr/bin/env python3
"""The tap-reuse convolution kernel (csrc/conv_halo.hip, TC_CONV_HALO) against the default routing of the same problem,
interleaved in one process, on every stride-1 3x3 / temporal convolution shape of the UNet at B = 2 (guided batch).
Prints per shape: default us and TF/s, halo us and the ratio, and whether the two results agree (max |diff| / scale)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tooncrafter_amd import ops
dev, BF = "cuda", torch.bfloat16
hip = ops.backend()

def timeit(fn, iters=20, reps=3):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / iters)
    return min(ts)

tot = {"default": 0.0, "halo": 0.0, "halo1g": 0.0, "tall": 0.0}

def conv(frames, h, w, cin, cout, tag, count, t3=False, emb=False):
    x = torch.randn(frames * h * w, cin, device=dev).to(BF)
    taps = 3 if t3 else 9
    wt = (torch.randn(cout, taps * cin, device=dev) * (taps * cin) ** -0.5).to(BF); b = torch.randn(cout, device=dev)
    res = torch.randn(frames * h * w, cout, device=dev).to(BF)
    geom = dict(kind="t3", frames=frames, t_len=16, cin=cin, h_out=h, w_out=w) if t3 else \
        dict(kind="3x3", frames=frames, cin=cin, h_in=h, w_in=w, h_out=h, w_out=w, stride=1, upsample=False)
    kw = dict(conv=geom, residual=res)
    if emb:
        kw = dict(conv=geom, row_bias=torch.randn(frames, cout, device=dev), row_div=h * w)
    fn = lambda: hip.gemm(x, wt, b, **kw)
    flops = 2.0 * frames * h * w * cout * taps * cin
    r = {"default": [], "halo": [], "halo1g": [], "tall": []}
    outs = {}
    for _ in range(2):
        # "tall" = 320-row patches where 20 rows (pixels) tile the image, the 160-row patches elsewhere (TC_CONV_HALO_TALL=2)
        # "halo" = 160-row patches, K split over two groups inside the block where the launch has at most 256 blocks
        # (TC_CONV_HALO_KSPLIT=1, level 2); "halo1g" = never split
        for name, v, tl, ks in (("default", "0", "0", "1"), ("halo", "2", "0", "1"), ("halo1g", "2", "0", "0"), ("tall", "2", "2", "1")):
            os.environ["TC_CONV_HALO"], os.environ["TC_CONV_HALO_TALL"], os.environ["TC_CONV_HALO_KSPLIT"] = v, tl, ks
            outs[name] = fn()
            r[name].append(timeit(fn))
    os.environ["TC_CONV_HALO"] = os.environ["TC_CONV_HALO_TALL"] = "0"; os.environ["TC_CONV_HALO_KSPLIT"] = "1"
    t = {k: min(v) * 1e3 for k, v in r.items()}
    d = (outs["halo"].float() - outs["default"].float()).abs().max().item() / max(outs["default"].float().abs().max().item(), 1e-9)
    same = torch.equal(outs["tall"], outs["halo1g"])
    for k in tot: tot[k] += count * t[k]
    print(f"{('convT3' if t3 else 'conv3x3') + ' ' + tag:16s} {cin:5d}->{cout:<5d} x{count:<3d} default {t['default']:7.1f} us {flops / t['default'] / 1e6:7.1f} TF/s | "
          f"halo {t['halo']:7.1f} us {flops / t['halo'] / 1e6:7.1f} TF/s x{t['default'] / t['halo']:5.3f} | one group {t['halo1g']:7.1f} | tall {t['tall']:7.1f} us x{t['default'] / t['tall']:5.3f} | "
          f"rel diff {d:.1e} tall==halo {same}", flush=True)

# (count per guided forward) -- ResBlock in_layers / out_layers convolutions and the four temporal convolutions of each
# TemporalConvBlock, lvdm/modules/networks/openaimodel3d.py:154,179,255-266; levels 0 / 1 / 2 (level 3 is 5 x 8: no patches)
conv(32, 40, 64, 320, 320, "L0", 7); conv(32, 40, 64, 320, 320, "L0 +emb", 2, emb=True)
conv(32, 40, 64, 640, 320, "L0 +emb", 2, emb=True); conv(32, 40, 64, 960, 320, "L0 +emb", 1, emb=True)
conv(32, 20, 32, 640, 640, "L1", 6); conv(32, 20, 32, 320, 640, "L1 +emb", 1, emb=True); conv(32, 20, 32, 640, 640, "L1 +emb", 1, emb=True)
conv(32, 20, 32, 1920, 640, "L1 +emb", 1, emb=True); conv(32, 20, 32, 1280, 640, "L1 +emb", 1, emb=True); conv(32, 20, 32, 960, 640, "L1 +emb", 1, emb=True)
conv(32, 10, 16, 1280, 1280, "L2", 6); conv(32, 10, 16, 640, 1280, "L2 +emb", 1, emb=True); conv(32, 10, 16, 1280, 1280, "L2 +emb", 1, emb=True)
conv(32, 10, 16, 2560, 1280, "L2 +emb", 2, emb=True); conv(32, 10, 16, 1920, 1280, "L2 +emb", 1, emb=True)
conv(32, 40, 64, 320, 320, "L0", 20, t3=True); conv(32, 20, 32, 640, 640, "L1", 20, t3=True); conv(32, 10, 16, 1280, 1280, "L2", 20, t3=True)
conv(32, 5, 8, 1280, 1280, "L3", 28, t3=True)
print(f"sum over one guided forward: default {tot['default'] / 1e3:.2f} ms, halo {tot['halo'] / 1e3:.2f} ms, tall where it tiles {tot['tall'] / 1e3:.2f} ms")
