#!/usr/bin/env python3
"""tile16 vs 128-tile with CACHE-COLD activations: the input rotates over buffers totalling > 256 MiB (Infinity Cache),
as in the model where every GEMM reads what the previous kernel just wrote, not the same tensor again."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tooncrafter_amd import ops
dev, BF = "cuda", torch.bfloat16
hip = ops.backend()

def timeit(fns, iters=24, reps=5):
    for f in fns: f()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters): fns[i % len(fns)]()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / iters)
    return sorted(ts)[len(ts) // 2]

def ab(fns, flops, tag):
    r = {}
    for _ in range(2):
        for mode in ("0", "2"):
            os.environ["TC_GEMM_TILE16"] = mode
            r.setdefault(mode, []).append(timeit(fns))
    t0, t2 = min(r["0"]), min(r["2"])
    print(f"{tag:34s} 128-tile {t0*1e3:7.1f} us {flops/t0/1e9:7.1f} TF/s | tile16 {t2*1e3:7.1f} us {flops/t2/1e9:7.1f} TF/s | x{t0/t2:5.2f}", flush=True)

def conv(frames, h, w, cin, cout, nbuf, tag, epi="bias"):
    xs = [torch.randn(frames * h * w, cin, device=dev).to(BF) for _ in range(nbuf)]
    outs = [torch.empty(frames * h * w, cout, device=dev, dtype=BF) for _ in range(nbuf)]
    ress = [torch.randn(frames * h * w, cout, device=dev).to(BF) for _ in range(nbuf)]
    wt = (torch.randn(cout, 9 * cin, device=dev) * (9 * cin) ** -0.5).to(BF); b = torch.randn(cout, device=dev)
    rb = torch.randn(2, cout, device=dev)
    geom = dict(kind="3x3", frames=frames, cin=cin, h_in=h, w_in=w, h_out=h, w_out=w, stride=1, upsample=False)
    if epi == "bias":
        fns = [(lambda x=x, o=o: hip.gemm(x, wt, b, conv=geom, out=o)) for x, o in zip(xs, outs)]
    elif epi == "rowbias":
        fns = [(lambda x=x, o=o: hip.gemm(x, wt, b, conv=geom, out=o, row_bias=rb, row_div=frames * h * w // 2)) for x, o in zip(xs, outs)]
    else:
        fns = [(lambda x=x, o=o, r=r: hip.gemm(x, wt, b, conv=geom, out=o, residual=r)) for x, o, r in zip(xs, outs, ress)]
    ab(fns, 2.0 * frames * h * w * cout * 9 * cin, f"conv {tag} {cin}->{cout} nbuf={nbuf} {epi}")

for nbuf in (1, 8):
    conv(32, 40, 64, 320, 320, nbuf, "L0")
    conv(32, 40, 64, 320, 320, nbuf, "L0", "rowbias")
    conv(32, 40, 64, 320, 320, nbuf, "L0", "residual")
    conv(32, 20, 32, 640, 640, nbuf * 2, "L1")
