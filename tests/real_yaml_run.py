#!/usr/bin/env python3
"""Helper of tests/test_gpu_real_yaml.py (run in a subprocess so that TC_BINDING selects the operator binding before
the backend exists): the reference's REAL configuration on the HIP backend, end to end.

  1. `tooncrafter_amd.dropin.install()` publishes the mirror under the reference's dotted paths;
  2. `utils.utils.instantiate_from_config` (the name inference.py:16 imports) builds
     `lvdm.models.ddpm3d.LatentVisualDiffusion` from tests/golden/inference_512_v1.0.model.yaml -- the values of the
     reference's configs/inference_512_v1.0.yaml, every `target:` path and parameter name as the reference has them
     (OpenCLIP ViT-H/14 towers and the Resampler included);
  3. a synthetic checkpoint in the reference's format ({"state_dict": ...}) is loaded with `strict=True`
     (inference.py:28-32);
  4. `image_guided_synthesis` (inference.py:180-277, as tooncrafter_amd/clip.py exposes it) runs 2 DDIM steps at
     320 x 512 x 16 frames with CFG 7.5 from a start / end frame pair, exactly as inference.py:324-342 calls it.
Prints one JSON line.
"""
import json
import os
import sys
import time

import torch
import yaml

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)


def main():
    t0 = time.time()
    from tooncrafter_amd import dropin, ops, synth
    dropin.install()
    from utils.utils import instantiate_from_config                      # the mirror, under the reference's name
    import lvdm.models.ddpm3d as ddpm3d
    assert ddpm3d.__name__ == "tooncrafter_amd.lvdm.ddpm3d"
    be = ops.backend()
    assert be.name == "hip"
    with open(os.path.join(ROOT, "tests", "golden", "inference_512_v1.0.model.yaml")) as f:
        cfg = yaml.safe_load(f)
    cfg["model"]["params"]["unet_config"]["params"]["use_checkpoint"] = False       # inference.py:286
    # construct on the CPU like the script does (inference.py:287), default initialisers skipped: every parameter is
    # overwritten by the checkpoint below
    import torch.nn.init as init
    saved = {n: getattr(init, n) for n in ("kaiming_uniform_", "uniform_", "normal_", "trunc_normal_", "xavier_uniform_")}
    for n in saved:
        setattr(init, n, lambda t, *a, **k: t)
    try:
        model = instantiate_from_config(cfg["model"])
    finally:
        for n, fn in saved.items():
            setattr(init, n, fn)
    model = model.cuda(0)                                                # inference.py:288
    model.perframe_ae = True                                             # inference.py:289
    t_build = time.time() - t0
    sd = {}
    for k, v in model.state_dict().items():
        if v.dtype.is_floating_point and v.dim() > 0 and k.split(".")[0] not in ("betas",) and not k.startswith(
                ("alphas_", "sqrt_", "log_", "posterior_", "scale_arr", "lvlb", "scale_factor")):
            sd[k] = synth.synth_tensor(k, tuple(v.shape), 77, v.device)
        else:
            sd[k] = v.clone()                                            # schedule buffers, scalars
    n_param_keys = sum(1 for k in sd)
    missing, unexpected = model.load_state_dict({"state_dict": sd}["state_dict"], strict=True)
    assert not missing and not unexpected
    del sd
    model.eval()
    t_load = time.time() - t0 - t_build

    from tooncrafter_amd.clip import image_guided_synthesis
    g = torch.Generator().manual_seed(5)
    fa, fb = (torch.rand(1, 3, 1, 320, 512, generator=g) * 2 - 1 for _ in range(2))
    videos = torch.cat([fa.repeat(1, 1, 8, 1, 1), fb.repeat(1, 1, 8, 1, 1)], dim=2).cuda(0)      # load_data_prompts, interp
    torch.manual_seed(123)
    with torch.no_grad():
        out = image_guided_synthesis(model, [""], videos, [1, 4, 16, 40, 64], n_samples=1, ddim_steps=2, ddim_eta=1.0,
                                     unconditional_guidance_scale=7.5, cfg_img=None, fs=10, text_input=False,
                                     multiple_cond_cfg=False, loop=False, interp=True,
                                     timestep_spacing="uniform_trailing", guidance_rescale=0.7)
    torch.cuda.synchronize()
    print(json.dumps({"binding": type(be).__name__, "shape": list(out.shape), "finite": bool(torch.isfinite(out).all()),
                      "std": float(out.float().std()), "state_dict_keys": n_param_keys,
                      "params_B": round(sum(p.numel() for p in model.parameters()) / 1e9, 3),
                      "build_s": round(t_build, 1), "load_s": round(t_load, 1), "total_s": round(time.time() - t0, 1)}))


if __name__ == "__main__":
    main()
