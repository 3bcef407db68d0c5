"""A/B of two builds of libtooncrafter_hip.so on the GEGLU layers (usage: geglu_pad_ab.py [path/to/other/lib.so]):
one subprocess per library, three interleaved rounds."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch
    import tooncrafter_amd._lib as L
    if sys.argv[2] != "-":
        L.LIB_PATH = sys.argv[2]
    from tooncrafter_amd import ops
    from tooncrafter_amd._lib import ACT_GEGLU
    hip = ops.backend()
    for m, n, k, tag in [(81920, 2560, 320, "L0"), (20480, 5120, 640, "L1"), (5120, 10240, 1280, "L2 (wide kernel)")]:
        a = torch.randn(m, k, device="cuda").to(torch.bfloat16)
        w = (torch.randn(n, k, device="cuda") * k ** -0.5).to(torch.bfloat16)
        b = torch.randn(n, device="cuda")
        for _ in range(3):
            hip.gemm(a, w, b, act=ACT_GEGLU)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            hip.gemm(a, w, b, act=ACT_GEGLU)
        e1.record()
        torch.cuda.synchronize()
        print(f"  GEGLU {tag}: {e0.elapsed_time(e1) / 30 * 1e3:7.1f} us", flush=True)
    sys.exit(0)
base = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tooncrafter_amd", "build", "base", "libtooncrafter_hip.so")
for rnd in range(3):
    for name, path in (("base (row stride 128)", base), ("padded GEGLU tile", "-")):
        print(f"{name}, round {rnd}:", flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", path], check=False)
