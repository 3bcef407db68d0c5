#!/usr/bin/env python3
"""Why do the short-K projections cost 35-45 % more inside a UNet forward than in the microbenchmarks
(profiles/r02_clip_kernel_stats.txt: 54.9 / 41.3 / 36.8 us at levels 0 / 1 / 2 against 41.2 / 30.0 / 25.3 us in
profiles/r02_tile16_ab.txt)?  In the model every layer has its OWN weights (2.9 GB per forward: always first touch
from HBM) and reads activations another kernel just wrote; the microbenchmarks reuse one weight matrix.  This probe
separates the two: activations hot / rotated past the 256 MB Infinity Cache, weights hot / rotated likewise.

    python scripts/cold_w_probe.py > gpurun_out/cold_w_probe.txt
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tooncrafter_amd import ops  # noqa: E402

DEV, BF = "cuda", torch.bfloat16
hip = ops.backend()


def timeit(fn, n_sets, iters=48, rounds=5):
    for i in range(min(n_sets, 4)):
        fn(i)
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e3)
    return sorted(ts)[len(ts) // 2]


def case(tag, m, n, k, res=True):
    a_bytes, w_bytes = m * k * 2, n * k * 2
    na = max(2, (320 << 20) // (a_bytes * (3 if res else 2)) + 1)     # A + C (+ residual) sets > 256 MiB together
    nw = max(2, (320 << 20) // w_bytes + 1)
    nw = min(nw, 256)
    a = [torch.randn(m, k, device=DEV).to(BF) for _ in range(na)]
    r = [torch.randn(m, n, device=DEV).to(BF) for _ in range(na)] if res else None
    o = [torch.empty(m, n, device=DEV, dtype=BF) for _ in range(na)]
    w = [(torch.randn(n, k, device=DEV) * k ** -0.5).to(BF) for _ in range(nw)]
    b = torch.randn(n, device=DEV)

    def run(ia, iw):
        hip.gemm(a[ia], w[iw], b, residual=r[ia] if res else None, out=o[ia])

    fl = 2.0 * m * n * k
    rows = []
    for name, fa, fw in (("A hot,  W hot ", lambda i: 0, lambda i: 0),
                         ("A cold, W hot ", lambda i: i % na, lambda i: 0),
                         ("A hot,  W cold", lambda i: 0, lambda i: i % nw),
                         ("A cold, W cold", lambda i: i % na, lambda i: i % nw)):
        t = timeit(lambda i: run(fa(i), fw(i)), max(na, nw))
        rows.append(f"{name} {t:7.1f} us {fl / t / 1e6:7.1f} TF/s")
    print(f"{tag:34s} (A sets {na}, W sets {nw})  " + " | ".join(rows), flush=True)


if __name__ == "__main__":
    print(f"# {torch.cuda.get_device_name(0)}; tc_gemm_bf16 linear + bias + residual; 'cold' = operand rotated over > 320 MiB")
    case("L0 proj 81920x320x320", 81920, 320, 320)
    case("L0 ff2 81920x320x1280", 81920, 320, 1280)
    case("L1 proj 20480x640x640", 20480, 640, 640)
    case("L1 qkv 20480x1920x640", 20480, 1920, 640, res=False)
    case("L1 ff2 20480x640x2560", 20480, 640, 2560)
    case("L2 proj 5120x1280x1280", 5120, 1280, 1280)
    case("L2 qkv 5120x3840x1280", 5120, 3840, 1280, res=False)
    case("L2 ff2 5120x1280x5120", 5120, 1280, 5120)
    case("L3 proj 1280x1280x1280", 1280, 1280, 1280)
