"""The reference's UNMODIFIED entry point, scripts/evaluation/inference.py, executed through
`tooncrafter_amd.dropin` against this package's mirror (the drop-in boundary):

    python -m tooncrafter_amd.dropin <reference>/scripts/evaluation/inference.py --config <yaml> --ckpt_path ... \
        --prompt_dir ... --savedir ... --interp --perframe_ae ...

with a YAML of exactly the structure of configs/inference_512_v1.0.yaml (same `target:` class paths, same
parameter names) at tiny widths, a checkpoint saved in the reference's format and loaded by ITS
`load_model_checkpoint` (`load_state_dict(strict=True)`), its own data loader (PIL + torchvision transforms), its
`image_guided_synthesis` and its `save_results_seperate`.  Runs on the CPU with the emulated operator contract
(tests/emu_ops.py): what is checked is the plumbing -- every import, constructor kwarg, state-dict key, call
signature and tensor shape the script relies on -- not kernel numerics (tests/test_gpu_*.py do that).

Three CPU-only accommodations, all in THIS file, none in the script: `Module.cuda()` / `Tensor.to("cuda")` are
redirected to the CPU (there is no GPU here), and the OpenCLIP towers are registered at a tiny geometry
(`arch: tiny-test`, a constructor kwarg the reference classes have too).
"""
import os
import sys

import numpy as np
import pytest
import torch
import yaml

from conftest import TINY_DD_CFG, TINY_UNET_CFG
from emu_ops import EmuOps
from tooncrafter_amd import ops, synth

REF_SCRIPT = "/root/reference/scripts/evaluation/inference.py"


def _tiny_yaml():
    unet = dict(TINY_UNET_CFG, use_checkpoint=True)             # the script flips this to False itself (:286)
    return {"model": {"target": "lvdm.models.ddpm3d.LatentVisualDiffusion", "params": {
        "rescale_betas_zero_snr": True, "parameterization": "v", "linear_start": 0.00085, "linear_end": 0.012,
        "num_timesteps_cond": 1, "timesteps": 1000, "first_stage_key": "video", "cond_stage_key": "caption",
        "cond_stage_trainable": False, "conditioning_key": "hybrid", "image_size": [8, 8], "channels": 4,
        "scale_by_std": False, "scale_factor": 0.18215, "use_ema": False, "uncond_type": "empty_seq",
        "use_dynamic_rescale": True, "base_scale": 0.7, "fps_condition_type": "fps", "perframe_ae": True,
        "loop_video": True,
        "unet_config": {"target": "lvdm.modules.networks.openaimodel3d.UNetModel", "params": unet},
        "first_stage_config": {"target": "lvdm.models.autoencoder.AutoencoderKL_Dualref",
                               "params": {"embed_dim": 4, "monitor": "val/rec_loss", "ddconfig": dict(TINY_DD_CFG),
                                          "lossconfig": {"target": "torch.nn.Identity"}}},
        "cond_stage_config": {"target": "lvdm.modules.encoders.condition.FrozenOpenCLIPEmbedder",
                              "params": {"freeze": True, "layer": "penultimate", "arch": "tiny-test"}},
        "img_cond_stage_config": {"target": "lvdm.modules.encoders.condition.FrozenOpenCLIPImageEmbedderV2",
                                  "params": {"freeze": True, "arch": "tiny-test"}},
        "image_proj_stage_config": {"target": "lvdm.modules.encoders.resampler.Resampler",
                                    "params": {"dim": 96, "depth": 1, "dim_head": 64, "heads": 2, "num_queries": 16,
                                               "embedding_dim": 160, "output_dim": 96, "ff_mult": 2,
                                               "video_length": 4}}}}}


@pytest.mark.skipif(not os.path.exists(REF_SCRIPT), reason="the reference tree is only present in the build container")
def test_reference_inference_script_runs_unmodified_through_dropin(tmp_path):
    from PIL import Image

    from tooncrafter_amd import dropin
    from tooncrafter_amd.lvdm import openclip
    from tooncrafter_amd.utils import instantiate_from_config

    openclip.ARCH["tiny-test"] = dict(
        embed_dim=64, vision=dict(width=160, layers=2, heads=2, patch=14, image=42, mlp=320),
        text=dict(width=96, layers=3, heads=1, context=77, vocab=49408, mlp=192))
    cfg = _tiny_yaml()
    ypath = tmp_path / "inference_tiny.yaml"
    ypath.write_text(yaml.safe_dump(cfg))

    # checkpoint in the reference's format ({"state_dict": ...}), every parameter synthetic and non-trivial
    model = instantiate_from_config(cfg["model"])
    synth.fill_module_(model, seed=77)
    ckpt = tmp_path / "model.ckpt"
    torch.save({"state_dict": model.state_dict()}, ckpt)
    n_keys = len(model.state_dict())
    del model

    # prompt directory: one prompt line + the start / end frames (inference.py:62-104)
    pdir = tmp_path / "prompts"
    pdir.mkdir()
    (pdir / "test_prompts.txt").write_text("an anime scene\n")
    rng = np.random.default_rng(0)
    for name in ("clip_frame1.png", "clip_frame3.png"):
        Image.fromarray(rng.integers(0, 255, size=(80, 96, 3), dtype=np.uint8)).save(pdir / name)
    out = tmp_path / "results"

    saved_modules = dict(sys.modules)
    saved_argv, saved_path = list(sys.argv), list(sys.path)
    orig_to, orig_cuda_t, orig_cuda_m = torch.Tensor.to, torch.Tensor.cuda, torch.nn.Module.cuda

    def to_cpu(self, *a, **k):
        a = tuple("cpu" if isinstance(x, str) and x.startswith("cuda") else x for x in a)
        if isinstance(k.get("device"), str) and k["device"].startswith("cuda"):
            k["device"] = "cpu"
        return orig_to(self, *a, **k)
    prev = ops.set_backend(EmuOps(round_bf16=True))
    torch.Tensor.to, torch.Tensor.cuda = to_cpu, lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, device=None: self
    try:
        rc = dropin.main([REF_SCRIPT, "--config", str(ypath), "--ckpt_path", str(ckpt), "--prompt_dir", str(pdir),
                          "--savedir", str(out), "--height", "64", "--width", "64", "--video_length", "4",
                          "--ddim_steps", "2", "--ddim_eta", "1.0", "--unconditional_guidance_scale", "7.5",
                          "--frame_stride", "10", "--timestep_spacing", "uniform_trailing", "--guidance_rescale", "0.7",
                          "--perframe_ae", "--interp", "--seed", "123", "--bs", "1", "--n_samples", "1"])
        assert rc == 0
        # the script really ran on the mirror, not on the reference's own lvdm package
        assert sys.modules["lvdm.models.ddpm3d"].__name__ == "tooncrafter_amd.lvdm.ddpm3d"
        assert sys.modules["lvdm.models.samplers.ddim"].DDIMSampler.__module__ == "tooncrafter_amd.lvdm.ddim"
    finally:
        ops.set_backend(prev)
        torch.Tensor.to, torch.Tensor.cuda, torch.nn.Module.cuda = orig_to, orig_cuda_t, orig_cuda_m
        sys.argv[:], sys.path[:] = saved_argv, saved_path
        for k in [k for k in sys.modules if k not in saved_modules]:
            del sys.modules[k]
        sys.modules.update(saved_modules)
        openclip.ARCH.pop("tiny-test", None)

    files = sorted(os.listdir(out / "samples_separate"))
    assert files == ["clip_frame1_sample0.mp4"], files          # the name of inference.py:153: an H.264 .mp4 (tooncrafter_amd/mp4.py)
    from test_mp4_cpu import _decode
    clip = _decode(str(out / "samples_separate" / files[0]))
    assert (clip["w"], clip["h"], len(clip["frames"])) == (64, 64, 4)
    luma = np.stack([f[0] for f in clip["frames"]])
    assert luma.std() > 1.0, "decoded frames are constant"
    assert n_keys > 900                                          # UNet + AE + both towers + resampler + buffers
