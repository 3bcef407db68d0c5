"""Where the MXFP8 error of one full-size UNet forward comes from: rel-L2 against the committed fp32 oracle golden
with the fp8 routing restricted to layer classes (TC_FP8 modes and thresholds), next to the forward's GPU time."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import fullsize_cases as fc  # noqa: E402
from conftest import rel_l2  # noqa: E402
from tooncrafter_amd import ops, synth  # noqa: E402
from tooncrafter_amd.utils import instantiate_from_config  # noqa: E402

DEV = "cuda"
with torch.device("meta"):
    model = instantiate_from_config(dict(target="lvdm.models.ddpm3d.LatentVisualDiffusion", params=bench.MODEL_PARAMS))
model = model.to_empty(device=DEV).eval()
with torch.no_grad():
    for name, p in model.named_parameters():
        p.copy_(synth.synth_tensor(name, tuple(p.shape), 1234, "cpu"))
un = model.model.diffusion_model
golden = dict(np.load(fc.GOLDEN_FILE))
inp = fc.inputs()
ref = torch.from_numpy(golden["unet_y"])
be = ops.backend()
ts = torch.tensor([fc.UNET_T], device=DEV)
kw = dict(context=inp["cond"].to(DEV), fs=inp["fs"].to(DEV))
parts = [inp["x_T"].to(DEV), inp["c_concat"].to(DEV)]


def run(label, mode, **attrs):
    saved = {k: getattr(be, k) for k in attrs}
    be.fp8 = mode
    for k, v in attrs.items():
        setattr(be, k, v)
    c0 = dict(be.fp8_calls)
    with torch.no_grad():
        y = un(None, ts, x_parts=parts, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            un(None, ts, x_parts=parts, **kw)
        e1.record()
        torch.cuda.synchronize()
    be.fp8 = None
    for k, v in saved.items():
        setattr(be, k, v)
    n = be.fp8_calls["mx"] - c0["mx"]
    print(f"{label:58s} mx launches {n // 4:4d}  rel-L2 vs fp32 oracle {rel_l2(y.cpu(), ref):.3e}  B=1 forward {e0.elapsed_time(e1) / 3:6.2f} ms", flush=True)


run("bf16 (no fp8)", None)
run("TC_FP8=all", "all")
run("convolutions only", "conv")
run("TC_FP8=1: linear, K >= 640, N >= 1280, N >= 2 K", "linear")
run("linear, K >= 640, N >= 1280 (no N >= 2 K rule)", "linear", fp8_n_over_k=0.0)
run("convolutions with cin >= 640 only", "conv", fp8_min_cin=640)
run("convolutions with cin <= 640 only", "conv", fp8_max_cin=640)
run("convolutions except the input convolution (cin >= 320)", "conv", fp8_min_cin=320)
run("3x3 convolutions only, cin >= 320", "conv3", fp8_min_cin=320)
run("temporal convolutions only", "convt")
run("all, convolutions cin >= 320", "all", fp8_min_cin=320)
run("every eligible linear (K >= 320, any N)", "linear", fp8_min_k=320, fp8_min_n=0, fp8_n_over_k=0.0)
run("wide-N linear incl. level 0 (K >= 320, N >= 960)", "linear", fp8_min_k=320, fp8_min_n=960)
run("wide-N linear incl. level 0 (K >= 320, N >= 1280)", "linear", fp8_min_k=320, fp8_min_n=1280)
run("linear default, LayerNorm fusion off", "linear", fp8_fuse_ln=False)
