"""Isolated timing of the MXFP8 GEMM (tc_gemm_mxfp8) against the bf16 kernels on UNet / decoder shapes:
bf16 GEMM, fp8 GEMM alone (operands already quantised), activation quantiser alone, and quantiser + fp8 GEMM
(what TC_FP8=1 pays per call).  Usage: python scripts/mx_bench.py [reps]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tooncrafter_amd import _lib  # noqa: E402
from tooncrafter_amd.ops import HipOps  # noqa: E402

BF16 = torch.bfloat16
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def timed(fn, reps=REPS):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3     # us


def main():
    h = HipOps()
    h8 = HipOps()
    h8.fp8, h8.fp8_min_k, h8.fp8_min_m, h8.fp8_min_n, h8.fp8_max_cin, h8.fp8_n_over_k = "all", 0, 1, 0, 1 << 20, 0.0
    cases = [
        ("L0 conv3x3 320->320 (B=2)", dict(frames=32, h=40, w=64, cin=320, cout=320)),
        ("L1 conv3x3 640->640", dict(frames=32, h=20, w=32, cin=640, cout=640)),
        ("L2 conv3x3 1280->1280", dict(frames=32, h=10, w=16, cin=1280, cout=1280)),
        ("L1 conv3x3 1920->640 (skip concat)", dict(frames=32, h=20, w=32, cin=1920, cout=640)),
        ("dec conv3x3 128->128 (4 frames)", dict(frames=4, h=320, w=512, cin=128, cout=128)),
        ("L0 ff2 81920x320x1280", dict(m=81920, n=320, k=1280)),
        ("L0 qkv 81920x960x320", dict(m=81920, n=960, k=320)),
        ("L1 qkv 20480x1920x640", dict(m=20480, n=1920, k=640)),
        ("L1 ff2 20480x640x2560", dict(m=20480, n=640, k=2560)),
        ("L2 ff2 5120x1280x5120", dict(m=5120, n=1280, k=5120)),
    ]
    print(f"{'case':38s} {'bf16 us':>9s} {'TF/s':>7s} {'mx us':>9s} {'TF/s':>7s} {'quant us':>9s} {'q+mx us':>9s}  speedup(kernel / with quant)")
    for name, c in cases:
        if "cin" in c:
            rows = c["frames"] * c["h"] * c["w"]
            a = (torch.randn(rows, c["cin"], device="cuda")).to(BF16)
            w = (torch.randn(c["cout"], 9 * c["cin"], device="cuda") * (9 * c["cin"]) ** -0.5).to(BF16)
            geom = dict(kind="3x3", frames=c["frames"], cin=c["cin"], h_in=c["h"], w_in=c["w"], h_out=c["h"], w_out=c["w"],
                        stride=1, upsample=False)
            m, n, k, kc = rows, c["cout"], 9 * c["cin"], c["cin"]
        else:
            m, n, k = c["m"], c["n"], c["k"]
            a = torch.randn(m, k, device="cuda").to(BF16)
            w = (torch.randn(n, k, device="cuda") * k ** -0.5).to(BF16)
            geom, kc = None, k
        bias = torch.randn(n, device="cuda")
        out = torch.empty((m, n), dtype=BF16, device="cuda")
        flops = 2.0 * m * n * k
        t_bf = timed(lambda: h.gemm(a, w, bias, conv=geom, out=out))
        t_all = timed(lambda: h8.gemm(a, w, bias, conv=geom, out=out))
        t_q = timed(lambda: h8.quant_mxfp8(a, kc))
        # GEMM alone: prebuilt parameter block over quantised operands
        aq, asc = h8.quant_mxfp8(a, kc)
        wq, wsc = h8._weight_mx(w)
        orig = h8.quant_mxfp8
        h8.quant_mxfp8 = lambda x, kk=None: (aq, asc)
        t_mx = timed(lambda: h8.gemm(a, w, bias, conv=geom, out=out))
        h8.quant_mxfp8 = orig
        print(f"{name:38s} {t_bf:9.1f} {flops / t_bf * 1e-6:7.0f} {t_mx:9.1f} {flops / t_mx * 1e-6:7.0f} {t_q:9.1f} {t_all:9.1f}"
              f"  {t_bf / t_mx:5.2f}x / {t_bf / t_all:5.2f}x", flush=True)


if __name__ == "__main__":
    main()
