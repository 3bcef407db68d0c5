#!/usr/bin/env python3
"""Where should a weight stream ride?    A side stream reads cold buffers (a torch reduction over
rotating 28 MB tensors, back to back) while the main stream runs ONE kind of launch, sustained; the main launches are timed
with and without the side stream, and the side stream's rate is what it moved over the window.
    hosts: a level-3 3x3 convolution (GEMM: compute / L2->LDS bound), a level-3 temporal convolution, a level-2 projection,
           the level-3 clip-wide GroupNorm (64 one-pass blocks), the level-2 per-frame GroupNorm, a level-3 LayerNorm
usage: python scripts/stream_host_probe.py > gpurun_out/TAG/stream_host_probe.txt"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tooncrafter_amd import ops

dev = torch.device("cuda:0")
lib = ctypes.CDLL(os.path.join(ROOT, "scripts", "bin", "libclock_probe.so"))
lib.clk_probe.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
lib.clk_probe.restype = ctypes.c_int
slots = torch.zeros(1024, 2, dtype=torch.int64, device=dev)
_next = [0]
hip = ops.backend()
hip.prefetch_on = False
BF = torch.bfloat16
side = torch.cuda.Stream()
cold = [torch.randn(1280, 11520, device=dev).to(BF) for _ in range(28)]          # 28 x 28 MB = 790 MB: every read comes from HBM


def probe():
    i = _next[0]
    _next[0] += 1
    assert lib.clk_probe(slots[i].data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    return i


def graph_of(fn, n):
    fn(); fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    return g


def side_graph():
    with torch.cuda.stream(side):
        for t in cold[:2]:
            t.view(torch.int16).amax()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for t in cold:
                t.view(torch.int16).amax()
    return g


SG = side_graph()
with torch.cuda.stream(side):                      # the stream alone: its rate with the chip to itself
    SG.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        SG.replay()
    torch.cuda.synchronize()
    alone_rate = 5 * 28 * cold[0].numel() * 2 / (time.perf_counter() - t0) / 1e12
print(f"# side stream alone: {alone_rate:.2f} TB/s of cold reads (torch amax over 28 MB tensors, back to back)")


def host(tag, fn, est_us):
    n = max(int(2000 / est_us), 8)                  # ~2 ms per graph
    g = graph_of(fn, n)
    reps = max(int(60e3 / (est_us * n)), 3)

    def timed(with_side):
        torch.cuda.synchronize()
        if with_side:
            with torch.cuda.stream(side):
                for _ in range(400):                # far more than the window needs; drained below
                    SG.replay()
            time.sleep(0.002)
        for _ in range(reps // 2 + 1):
            g.replay()
        a = probe()
        for _ in range(reps):
            g.replay()
        b = probe()
        torch.cuda.current_stream().synchronize()
        s = slots.cpu()
        return float(s[b, 1] - s[a, 1]) / 100.0 / (reps * n)
    base = timed(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    withs = timed(True)
    win = time.perf_counter() - t0
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    base2 = timed(False)
    b = (base + base2) / 2
    print(f"{tag:40s} alone {b:7.1f} us | beside the stream {withs:7.1f} us (x{withs / b:.3f}) | the 400 side replays took {tot * 1e3:7.1f} ms in all "
          f"({400 * 28 * cold[0].numel() * 2 / tot / 1e12:.2f} TB/s incl. the tail after the window of {win * 1e3:.0f} ms)", flush=True)


def c3(frames, h, w_, cin):
    return dict(kind="3x3", frames=frames, cin=cin, h_in=h, w_in=w_, h_out=h, w_out=w_, stride=1, upsample=False)


def t3(frames, h, w_, cin):
    return dict(kind="t3", frames=frames, t_len=16, cin=cin, h_in=h, w_in=w_, h_out=h, w_out=w_)


with torch.no_grad():
    a3 = torch.randn(1280, 1280, device=dev).to(BF)
    a2 = torch.randn(5120, 1280, device=dev).to(BF)
    w33 = (torch.randn(1280, 11520, device=dev) * 0.01).to(BF)
    wt3 = (torch.randn(1280, 3840, device=dev) * 0.02).to(BF)
    wl = (torch.randn(1280, 1280, device=dev) * 0.03).to(BF)
    gam, bet = torch.ones(1280, device=dev), torch.zeros(1280, device=dev)
    host("level-3 3x3 convolution (GEMM)", lambda: hip.gemm(a3, w33, conv=c3(32, 5, 8, 1280)), 52)
    host("level-3 temporal convolution (GEMM)", lambda: hip.gemm(a3, wt3, conv=t3(32, 5, 8, 1280)), 26)
    host("level-2 projection (GEMM)", lambda: hip.gemm(a2, wl), 22)
    host("level-2 3x3 convolution (halo GEMM)", lambda: hip.gemm(a2, w33, conv=c3(32, 10, 16, 1280)), 125)
    host("level-3 clip-wide GroupNorm (64 blocks)", lambda: hip.groupnorm(a3, gam, bet, samples=2, rows=640, eps=1e-5, silu=True), 12)
    host("level-2 per-frame GroupNorm", lambda: hip.groupnorm(a2, gam, bet, samples=32, rows=160, eps=1e-5, silu=True), 10)
    host("level-3 LayerNorm", lambda: hip.layernorm(a3, gam, bet), 8)
