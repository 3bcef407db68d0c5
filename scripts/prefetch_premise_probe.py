#!/usr/bin/env python3
"""Premise of a weight prefetch: if ANOTHER kernel reads a GEMM's weights one launch earlier (so that they sit in the 256 MB
Infinity Cache, not in the consumer's L2), how much of the cold-weight penalty of scripts/cold_operand_probe.py goes away?
Per shape, as hipGraphs of L (touch, GEMM) pairs over weight copies that rotate through >= 768 MB:
    cold     touch(unrelated buffer of W's size, rotating)  ; GEMM(a, W[i])      -- W[i] comes from HBM
    touched  touch(W[i])                                    ; GEMM(a, W[i])      -- W[i] was read by the launch in front
    touched2 touch(W[i+1]) ; GEMM(a, W[i])  (the touch runs TWO launches ahead of its consumer)
    warm     touch(unrelated, rotating)                     ; GEMM(a, W[0])      -- the per-shape loop's weights
touch = a torch reduction over the tensor (an HBM-bound read); its cost is in every arm.
usage: python scripts/prefetch_premise_probe.py > gpurun_out/TAG/prefetch_premise_probe.txt"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tooncrafter_amd import ops

dev = torch.device("cuda:0")
lib = ctypes.CDLL(os.path.join(ROOT, "scripts", "bin", "libclock_probe.so"))
lib.clk_probe.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
lib.clk_probe.restype = ctypes.c_int
slots = torch.zeros(8192, 2, dtype=torch.int64, device=dev)
_next = [0]
hip = ops.backend()
BF = torch.bfloat16
COLD_BYTES = 768 << 20


def probe():
    i = _next[0]
    _next[0] += 1
    assert lib.clk_probe(slots[i].data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    return i


def graph_of(launches):
    for f in launches[:2]:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for f in launches:
            f()
    return g


def run(graphs, per_graph, reps):
    rec = {k: [] for k in graphs}
    for _ in range(2):
        for k, g in graphs.items():
            for _ in range(max(reps // 2, 1)):
                g.replay()
            a = probe()
            for _ in range(reps):
                g.replay()
            rec[k].append((a, probe()))
    torch.cuda.synchronize()
    s = slots.cpu()
    return {k: sum(float(s[b, 1] - s[a, 1]) / 100.0 / (reps * per_graph) for a, b in v) / len(v) for k, v in rec.items()}


def shape(tag, m, n, k, est_us, conv=None):
    a = torch.randn(m if conv is None else conv["frames"] * conv["h_in"] * conv["w_in"], k if conv is None else conv["cin"], device=dev).to(BF)
    w = (torch.randn(n, k, device=dev) * k ** -0.5).to(BF)
    r = min(max(int(COLD_BYTES // (w.numel() * 2)) + 1, 2), 192)
    ww = [w.clone() for _ in range(r)]
    tt = [w.clone() for _ in range(r)]                     # unrelated buffers of the same size
    kw = {"conv": conv} if conv is not None else {}
    L = len(ww)
    touch = lambda t: (lambda: t.view(torch.int16).amax())
    gemm = lambda W: (lambda: hip.gemm(a, W, **kw))
    seq = lambda f: [x for i in range(L) for x in f(i)]
    graphs = {"cold": graph_of(seq(lambda i: (touch(tt[i]), gemm(ww[i])))),
              "touched": graph_of(seq(lambda i: (touch(ww[i]), gemm(ww[i])))),
              "touched2": graph_of(seq(lambda i: (touch(ww[(i + 1) % L]), gemm(ww[i])))),
              "warm": graph_of(seq(lambda i: (touch(tt[i]), gemm(ww[0])))),
              "touch only": graph_of(seq(lambda i: (touch(tt[i]),)))}
    reps = max(int(100e3 / ((est_us + 15) * L)), 2)
    t = run(graphs, L, reps)
    to = t["touch only"]
    c, w1, w2, wm = t["cold"] - to, t["touched"] - to, t["touched2"] - to, t["warm"] - to
    rec = (c - w1) / (c - wm) if c > wm else float("nan")
    print(f"{tag:30s} W {w.numel() * 2 / 2 ** 20:5.1f} MB x {L:3d} | touch {to:6.1f} us | GEMM: cold {c:6.1f}  touched {w1:6.1f}  touched 2 ahead {w2:6.1f}  warm {wm:6.1f} us | "
          f"recovered {100 * rec:4.0f} % / {100 * (c - w2) / (c - wm) if c > wm else float('nan'):4.0f} % of the cold-weight penalty ({100 * (c / wm - 1):.1f} %)", flush=True)


def c3(frames, h, w_, cin):
    return dict(kind="3x3", frames=frames, cin=cin, h_in=h, w_in=w_, h_out=h, w_out=w_, stride=1, upsample=False)


def t3(frames, h, w_, cin):
    return dict(kind="t3", frames=frames, t_len=16, cin=cin, h_in=h, w_in=w_, h_out=h, w_out=w_)


print("# per (touch, GEMM) pair, sustained graphs; GEMM time = pair - touch-only")
with torch.no_grad():
    shape("L3 3x3 1280->1280", 1280, 1280, 11520, 62, conv=c3(32, 5, 8, 1280))
    shape("L3 t3 1280->1280", 1280, 1280, 3840, 29, conv=t3(32, 5, 8, 1280))
    shape("L3 qkv 1280x3840x1280", 1280, 3840, 1280, 25)
    shape("L3 proj 1280x1280x1280", 1280, 1280, 1280, 12)
    shape("L3 GEGLU-shape 1280x10240x1280", 1280, 10240, 1280, 42)
    shape("L2 3x3 1280->1280", 5120, 1280, 11520, 130, conv=c3(32, 10, 16, 1280))
    shape("L2 qkv 5120x3840x1280", 5120, 3840, 1280, 57)
    shape("L2 proj 5120x1280x1280", 5120, 1280, 1280, 23)
    shape("L2 t3 1280->1280", 5120, 1280, 3840, 54, conv=t3(32, 10, 16, 1280))
