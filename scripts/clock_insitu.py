#!/usr/bin/env python3
"""Engine clock IN SITU: is the guided UNet forward slower inside the clip than the sum of its kernels measured one shape
at a time because the chip clocks down under a sustained dense load, or for another reason?

Two in-stream probes (scripts/probes/clock_probe.hip: s_memtime = shader-clock counter, s_memrealtime = constant 100 MHz)
bracket a stretch of work on the launch stream: average engine clock = d(clock64) / d(wall_clock64) x 100 MHz, and the
stretch's duration comes from the same 100 MHz counter.  Regimes:
  forward, sustained : hipGraph replays of the B = 2 forward back to back (what a clip is: 50 of them)
  forward, bursts    : ONE replay, then the host sleeps -- the duty cycle of a per-shape microbenchmark at forward granularity
  shape, sustained / bursts : the same for single tc_gemm_bf16 shapes (bursts of 25 launches = what gemm_autotune.py times)
usage: python scripts/clock_insitu.py > gpurun_out/TAG/clock_insitu.txt"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from tooncrafter_amd import ops

dev = torch.device("cuda:0")
lib = ctypes.CDLL(os.path.join(ROOT, "scripts", "bin", "libclock_probe.so"))
lib.clk_probe.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
lib.clk_probe.restype = ctypes.c_int
NSLOT = 4096
slots = torch.zeros(NSLOT, 2, dtype=torch.int64, device=dev)
_next = [0]


def probe():
    i = _next[0]
    _next[0] += 1
    rc = lib.clk_probe(slots[i].data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
    return i


def span(i, j):
    """(engine MHz, milliseconds) between probes i and j (call after a synchronize)"""
    s = slots.cpu()
    dc, dw = int(s[j, 0] - s[i, 0]), int(s[j, 1] - s[i, 1])
    return (dc / dw * 100.0 if dw else float("nan")), dw / 100e3


def sustained(fn, warm, n):
    for _ in range(warm):
        fn()
    a = probe()
    for _ in range(n):
        fn()
    b = probe()
    torch.cuda.synchronize()
    mhz, ms = span(a, b)
    return mhz, ms / n


def bursts(fn, per_burst, n_bursts, pause):
    pairs = []
    for _ in range(n_bursts):
        torch.cuda.synchronize()
        time.sleep(pause)
        fn()                                   # the first launch after a pause pays the wake-up: not timed
        a = probe()
        for _ in range(per_burst):
            fn()
        pairs.append((a, probe()))
    torch.cuda.synchronize()
    r = [span(a, b) for a, b in pairs]
    r = r[len(r) // 2:]                        # the later bursts
    return sum(x[0] for x in r) / len(r), sum(x[1] for x in r) / len(r) / per_burst


def line(tag, sus, bur):
    print(f"{tag:44s} sustained {sus[1] * 1e3:9.1f} us @ {sus[0]:6.0f} MHz | bursts {bur[1] * 1e3:9.1f} us @ {bur[0]:6.0f} MHz | "
          f"time x{sus[1] / bur[1]:5.3f}  clock x{sus[0] / bur[0]:5.3f}", flush=True)


print(f"# clock64 / wall_clock64 probes; device {torch.cuda.get_device_name(0)}")
with torch.no_grad():
    model = bench.build_model(dev)
    inp = bench.make_inputs(dev, 7)
    fwd = bench.guided_forward(model, inp)
    fwd(); fwd()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fwd()
    # an idle probe pair: what the counters read with nothing running
    a = probe(); time.sleep(0.05); b = probe(); torch.cuda.synchronize()
    print(f"# idle 50 ms between two probes: {span(a, b)[0]:.0f} MHz (counter ratio with no load)")
    sus = sustained(g.replay, warm=15, n=30)
    bur = bursts(g.replay, per_burst=1, n_bursts=10, pause=0.25)
    line("guided UNet forward (B = 2), graph replay", sus, bur)
    sus2 = sustained(g.replay, warm=15, n=30)
    print(f"#   again, sustained: {sus2[1]:.2f} ms @ {sus2[0]:.0f} MHz")

    hip = ops.backend()
    BF = torch.bfloat16
    MIX = []                                   # (launch, ms sustained alone, ms in bursts alone)

    def lin(m, n, k, tag, **kw):
        a_ = torch.randn(m, k, device=dev).to(BF)
        w_ = (torch.randn(n, k, device=dev) * k ** -0.5).to(BF)
        fn = lambda: hip.gemm(a_, w_, **kw)
        fn(); torch.cuda.synchronize()
        per = max(int(0.4e-3 * 2.5e15 * 0.3 / (2.0 * m * n * k)), 8)       # ~0.4 ms per burst at 0.3 of the roof
        s_ = sustained(fn, warm=per * 200, n=per * 100)
        b_ = bursts(fn, per_burst=per, n_bursts=12, pause=0.05)
        line(f"linear {tag} {m}x{n}x{k} ({per}/burst)", s_, b_)
        MIX.append((fn, s_[1], b_[1]))

    def conv(frames, h, w, cin, cout, tag):
        x = torch.randn(frames * h * w, cin, device=dev).to(BF)
        wt = (torch.randn(cout, 9 * cin, device=dev) * (9 * cin) ** -0.5).to(BF)
        b = torch.randn(cout, device=dev)
        geom = dict(kind="3x3", frames=frames, cin=cin, h_in=h, w_in=w, h_out=h, w_out=w, stride=1, upsample=False)
        fn = lambda: hip.gemm(x, wt, b, conv=geom)
        fn(); torch.cuda.synchronize()
        per = max(int(0.4e-3 * 2.5e15 * 0.3 / (2.0 * frames * h * w * cout * 9 * cin)), 4)
        s_ = sustained(fn, warm=per * 200, n=per * 100)
        b_ = bursts(fn, per_burst=per, n_bursts=12, pause=0.05)
        line(f"conv3x3 {tag} {cin}->{cout} ({per}/burst)", s_, b_)
        MIX.append((fn, s_[1], b_[1]))

    lin(20480, 1920, 640, "L1 qkv")
    lin(5120, 1280, 1280, "L2 proj")
    lin(20480, 5120, 640, "L1 GEGLU-shape")
    lin(81920, 320, 320, "L0 proj (weight-stationary)")
    conv(32, 40, 64, 320, 320, "L0")
    conv(32, 20, 32, 640, 640, "L1")
    # GroupNorm: HBM-bound, should not care about the engine clock
    x = torch.randn(2 * 40960, 320, device=dev).to(BF)
    gam, bet = torch.ones(320, device=dev), torch.zeros(320, device=dev)
    fn = lambda: hip.groupnorm(x, gam, bet, samples=2, rows=40960, eps=1e-5, silu=True)
    fn(); torch.cuda.synchronize()
    line("groupnorm L0 clip-wide (12/burst)", sustained(fn, warm=2000, n=1200), bursts(fn, per_burst=12, n_bursts=12, pause=0.05))

    # the same launches ROUND-ROBIN (every kernel follows a different one: its code and operands were last touched five
    # launches ago), sustained: against the sum of the times each took alone
    def mix():
        for f, _, _ in MIX:
            f()
    m_ = sustained(mix, warm=300, n=300)
    alone_s, alone_b = sum(t for _, t, _ in MIX), sum(t for _, _, t in MIX)
    print(f"round-robin of the {len(MIX)} GEMM launches above, sustained: {m_[1] * 1e3:.1f} us per round @ {m_[0]:.0f} MHz | "
          f"sum of each alone: sustained {alone_s * 1e3:.1f} us, bursts {alone_b * 1e3:.1f} us | "
          f"x{m_[1] / alone_s:.3f} / x{m_[1] / alone_b:.3f}", flush=True)
