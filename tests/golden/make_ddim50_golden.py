#!/usr/bin/env python3
"""Writes tests/golden/ddim50_oracle.npz: the DDIM-50 trajectory of the fp32 oracle at the BASELINE shape (sampled
positions) and, per step, how far the SAME oracle under torch.autocast(bfloat16) lands from it (the noise floor;
measured over the full tensors).  Runs ON THE MI355X -- both oracles are PyTorch-eager restatements of the reference
(oracle/unet.py, oracle/sampler.py, oracle/decoder.py) and take ~200 s there, ~3 h on CPU cores -- from seeds only:
weights = synth.synth_tensor(name, shape, 1234), inputs = fullsize_cases.inputs(), noise draw i = seed 300 + i.

    python tests/golden/make_ddim50_golden.py        # on the GPU box; the file lands in gpurun_out/ as well
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import fullsize_cases as fc  # noqa: E402
import test_gpu_ddim50 as t50  # noqa: E402
from conftest import rel_l2  # noqa: E402
from tooncrafter_amd import synth  # noqa: E402
from tooncrafter_amd.utils import instantiate_from_config  # noqa: E402


def main():
    with torch.device("meta"):
        model = instantiate_from_config(dict(target="lvdm.models.ddpm3d.LatentVisualDiffusion", params=bench.MODEL_PARAMS))
    model = model.to_empty(device="cuda").eval()
    with torch.no_grad():
        for name, p in model.named_parameters():
            p.copy_(synth.synth_tensor(name, tuple(p.shape), 1234, "cpu"))
    inp = fc.inputs()
    runs = t50.oracle_trajectories(model, inp)
    ref, flo = runs["fp32"], runs["bf16"]
    ix0, ifin, ipix = t50.sample_positions(ref["x0s"][0].numel(), ref["final"].numel(), ref["pix"].numel())
    pick = lambda t, idx: t.reshape(-1)[idx.to(t.device)].float().cpu().numpy()
    out = dict(x0_fp32=np.stack([pick(x, ix0) for x in ref["x0s"]]), final_fp32=pick(ref["final"], ifin), pix_fp32=pick(ref["pix"], ipix),
               floor_x0=np.array([rel_l2(flo["x0s"][i], ref["x0s"][i]) for i in range(t50.S)], dtype=np.float64),
               floor_final=np.float64(rel_l2(flo["final"], ref["final"])), floor_pix=np.float64(rel_l2(flo["pix"], ref["pix"])),
               meta=np.array(f"{torch.cuda.get_device_name(0)}; torch {torch.__version__}; DDIM-{t50.S} CFG {fc.CFG} eta {fc.ETA} rescale {fc.RESCALE}"))
    for path in (t50.GOLDEN, os.path.join(ROOT, "gpurun_out", "ddim50_oracle.npz")):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        np.savez(path, **out)
    print("floors: x0", out["floor_x0"][:3], "...", out["floor_x0"][-3:], "final", out["floor_final"], "pixels", out["floor_pix"])


if __name__ == "__main__":
    main()
