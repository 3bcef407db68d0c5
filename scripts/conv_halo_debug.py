#!/usr/bin/env python3
"""Where is it wrong?  Localises a mismatch of the tap-reuse convolution kernel (csrc/conv_halo.hip) against the fp32
statement of the operator, for the FIRST GPU session with that kernel (it was written without one):

  python scripts/conv_halo_debug.py [3x3|t3] [frames h w cin n] [tall|ksplit]

1. the whole problem on random data: error by patch position (y, x), by frame / clip, by 16-column block;
2. ONE TAP at a time (all other taps' weights zero): a wrong tap offset / padding mask shows up as that tap alone;
3. ONE CHANNEL CHUNK at a time (all other input channels zero): a wrong refill / K order shows up as chunks >= 1 alone;
4. a one-hot input pixel: which output pixels it reaches (must be its 3 x 3 neighbourhood / its own pixel at frames t-1..t+1).
Everything under TC_CONV_HALO=2 (strict: the call FAILS instead of falling back)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402

from emu_ops import EmuOps  # noqa: E402
from tooncrafter_amd.ops import HipOps  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "3x3"
nums = [int(a) for a in sys.argv[2:7]] if len(sys.argv) >= 7 else ([2, 20, 32, 128, 160] if kind == "3x3" else [16, 4, 5, 128, 160])
flags = set(sys.argv[7:])
frames, h, w_, cin, n = nums
taps = 9 if kind == "3x3" else 3
os.environ["TC_CONV_HALO"] = "2"
os.environ["TC_CONV_HALO_TALL"] = "2" if "tall" in flags else "0"
os.environ["TC_CONV_HALO_KSPLIT"] = "2" if "ksplit" in flags else "0"
hip, emu = HipOps(), EmuOps(round_bf16=True)
dev, BF = "cuda", torch.bfloat16
m = frames * h * w_
conv = dict(kind="3x3", frames=frames, cin=cin, h_in=h, w_in=w_, h_out=h, w_out=w_, stride=1, upsample=False) if kind == "3x3" \
    else dict(kind="t3", frames=frames, t_len=16, cin=cin, h_out=h, w_out=w_)
g = torch.Generator().manual_seed(0)
x = torch.randn(m, cin, generator=g).to(BF).to(dev)
wt = (torch.randn(n, taps * cin, generator=g) * (taps * cin) ** -0.5).to(BF).to(dev)


def run(xx, ww):
    return hip.gemm(xx, ww, conv=conv).float(), emu.gemm(xx, ww, conv=conv).float()


def report(tag, got, ref):
    err = (got - ref).abs()
    scale = ref.abs().max().item() + 1e-30
    rel = ((got - ref).norm() / (ref.norm() + 1e-30)).item()
    print(f"{tag:34s} rel-L2 {rel:.3e}  max {err.max().item():.3e} (scale {scale:.3e})  wrong (> 2 ulp of scale): "
          f"{(err > scale * 2 ** -7).float().mean().item():.4f}")
    return err, scale


print(f"{kind} frames {frames} {h}x{w_} cin {cin} n {n} flags {sorted(flags)}")
got, ref = run(x, wt)
err, scale = report("1. random data", got, ref)
bad = err > scale * 2 ** -7
if bad.any():
    if kind == "3x3":
        e = bad.float().mean(1).view(frames, h, w_)
        print("   wrong fraction by image row   :", [round(v, 2) for v in e.mean((0, 2)).tolist()])
        print("   wrong fraction by image column:", [round(v, 2) for v in e.mean((0, 1)).tolist()])
        print("   wrong fraction by frame       :", [round(v, 2) for v in e.mean((1, 2)).tolist()])
    else:
        e = bad.float().mean(1).view(frames // 16, 16, h * w_)
        print("   wrong fraction by frame of clip:", [round(v, 2) for v in e.mean((0, 2)).tolist()])
        print("   wrong fraction by pixel        :", [round(v, 2) for v in e.mean((0, 1)).tolist()][:40])
    print("   wrong fraction by 16-column block:", [round(v, 2) for v in bad.float().mean(0).view(-1, 16).mean(1).tolist()])
for t in range(taps):
    w1 = torch.zeros_like(wt)
    w1[:, t * cin:(t + 1) * cin] = wt[:, t * cin:(t + 1) * cin]
    report(f"2. tap {t} alone", *run(x, w1))
if "gn" not in flags:
    for c in range(cin // 64):
        x1 = torch.zeros_like(x)
        x1[:, c * 64:(c + 1) * 64] = x[:, c * 64:(c + 1) * 64]
        report(f"3. channel chunk {c} alone", *run(x1, wt))
    for f, y, xx in ((0, 0, 0), (frames - 1, h - 1, w_ - 1), (frames // 2, h // 2, w_ // 2), (0, min(9, h - 1), min(15, w_ - 1)),
                     (0, min(10, h - 1), min(16, w_ - 1))):
        x1 = torch.zeros_like(x)
        x1[(f * h + y) * w_ + xx] = 1.0
        got, ref = run(x1, wt)
        hit_g = (got.abs().amax(1) > 0).view(frames, h, w_).nonzero().tolist()
        hit_r = (ref.abs().amax(1) > 0).view(frames, h, w_).nonzero().tolist()
        print(f"4. one-hot pixel (f {f}, y {y}, x {xx}): reaches {len(hit_g)} outputs, should reach {len(hit_r)}"
              + ("" if hit_g == hit_r else f"\n   got  {hit_g[:12]}\n   want {hit_r[:12]}"))
torch.cuda.synchronize()
