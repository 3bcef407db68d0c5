#!/usr/bin/env python3
"""Is a GEMM's A operand warm inside the forward?  There A is what the launch in front has just WRITTEN.  The cold-A arm of
scripts/cold_operand_probe.py (A last touched >= 768 MB ago: x1.18-1.43 on the short-K linear layers) is only an upper bound
unless written lines do not stay in the memory-side Infinity Cache.  Per shape, hipGraphs of L (producer, consumer) pairs; the
producer is a GEMM that writes A_i (explicit `out=` buffers rotating through >= 768 MB), the consumer the shape under test:
    fresh   P -> A_i     ; C reads A_i            (what the forward does)
    stale   P -> A_i     ; C reads A_(i + L/2)    (written ~L/2 pairs ago: >= 384 MB of writes in between)
    warm    P -> A_i     ; C reads one fixed A    (the per-shape loop)
    P only
consumer time = pair - P only.
usage: python scripts/fresh_activation_probe.py > gpurun_out/TAG/fresh_activation_probe.txt"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tooncrafter_amd.ops import HipOps

dev = torch.device("cuda:0")
lib = ctypes.CDLL(os.path.join(ROOT, "scripts", "bin", "libclock_probe.so"))
lib.clk_probe.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
lib.clk_probe.restype = ctypes.c_int
slots = torch.zeros(4096, 2, dtype=torch.int64, device=dev)
_next = [0]
hip = HipOps()                     # the ctypes binding: gemm(out=) with caller-owned outputs
BF = torch.bfloat16


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


def shape(tag, m, n, k, est_us):
    """consumer: [m, k] x [n, k]^T; producer: [m, 320] x [k, 320]^T -> A_i [m, k]"""
    src = torch.randn(m, 320, device=dev).to(BF)
    wp = (torch.randn(k, 320, device=dev) * 320 ** -0.5).to(BF)
    wc = (torch.randn(n, k, device=dev) * k ** -0.5).to(BF)
    L = min(max((768 << 20) // (m * k * 2) + 1, 8), 256)
    A = [torch.empty(m, k, device=dev, dtype=BF) for _ in range(L)]
    for a in A:
        hip.gemm(src, wp, out=a)
    a_fixed = A[0].clone()
    cout = torch.empty(m, n, device=dev, dtype=BF)
    P = lambda i: (lambda: hip.gemm(src, wp, out=A[i]))
    Cn = lambda a: (lambda: hip.gemm(a, wc, out=cout))
    seq = lambda f: [x for i in range(L) for x in f(i)]
    graphs = {"fresh": graph_of(seq(lambda i: (P(i), Cn(A[i])))),
              "stale": graph_of(seq(lambda i: (P(i), Cn(A[(i + L // 2) % L])))),
              "warm": graph_of(seq(lambda i: (P(i), Cn(a_fixed)))),
              "P only": graph_of(seq(lambda i: (P(i),)))}
    reps = max(int(100e3 / ((est_us * 2) * L)), 2)
    t = run(graphs, L, reps)
    po = t["P only"]
    f, s_, w = t["fresh"] - po, t["stale"] - po, t["warm"] - po
    print(f"{tag:28s} A {m * k * 2 / 2 ** 20:5.1f} MB x {L:3d} | producer {po:6.1f} us | consumer: fresh {f:6.1f}  stale {s_:6.1f}  warm {w:6.1f} us | "
          f"fresh / warm x{f / w:.3f}   stale / warm x{s_ / w:.3f}", flush=True)


print("# consumer GEMM time by where its A operand comes from (pair - producer only), sustained hipGraphs")
with torch.no_grad():
    shape("L1 proj 20480x640x640", 20480, 640, 640, 26)
    shape("L2 proj 5120x1280x1280", 5120, 1280, 1280, 23)
    shape("L2 ff2 5120x1280x5120", 5120, 1280, 5120, 71)
    shape("L3 proj 1280x1280x1280", 1280, 1280, 1280, 12)
    shape("L1 qkv 20480x1920x640", 20480, 1920, 640, 62)
    shape("L0 proj 81920x320x320", 81920, 320, 320, 33)
