"""Full-size (BASELINE.json configs[1]: 320-channel UNet, 16 frames, 40x64 latents, 320x512 pixels)
parity cases shared by the golden generator (tests/golden/make_fullsize_golden.py, runs the fp32 CPU
oracle -- minutes) and the GPU tests (tests/test_gpu_fullsize.py, runs the HIP path -- seconds).

Everything is a pure function of seeds: weights from tooncrafter_amd.synth (CPU generator, identical on
every box), inputs from synth_inputs / synth_ref_context, DDIM noise from seeded CPU generators.  The
golden file keeps the small tensors whole and SAMPLES the big ones at fixed pseudo-random positions
(`sample_idx`), so a few MB pin 7.9 M-pixel outputs.
"""
import json
import os

import torch

from conftest import FULL_DD_CFG, FULL_UNET_CFG, GOLDEN
from tooncrafter_amd import synth

GOLDEN_FILE = os.path.join(GOLDEN, "fullsize_oracle.npz")
T, H, W = 16, 40, 64
UNET_T = 601                       # timestep of the single-forward case
DDIM_STEPS, CFG, RESCALE, ETA = 3, 7.5, 0.7, 1.0
N_OUT, N_PROBE = 131072, 32768     # sampled positions per decoded clip / per decoder stage
PROBES = ("mid", "level3", "level2", "level1", "level0")
IDX14 = [i for i in range(T) if i not in (1, T - 2)]      # inference.py:264-267


def sample_idx(numel: int, n: int, seed: int) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, numel, (min(n, numel),), generator=g)


def full_state_dict(prefixes):
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        man = json.load(f)["full"]
    return synth.synth_state_dict({k: tuple(v) for k, v in man.items() if k.startswith(prefixes)}, seed=1234)


def inputs():
    inp = synth.synth_inputs(1, T, H, W, seed=7)
    inp["refs"] = synth.synth_ref_context(1, H, W, ch=128, seed=107)
    g = torch.Generator().manual_seed(2025)
    inp["z_dec"] = torch.randn(1, 4, T, H, W, generator=g)
    inp["noises"] = [torch.randn(1, 4, T, H, W, generator=torch.Generator().manual_seed(300 + i))
                     for i in range(DDIM_STEPS)]
    return inp


def nchw_flat_from_rows(rows: torch.Tensor, frames: int, h: int, w: int) -> torch.Tensor:
    """HIP activation rows [(f h w), C] -> flat view in the oracle's (f, C, h, w) order."""
    c = rows.shape[1]
    return rows.reshape(frames, h, w, c).permute(0, 3, 1, 2).reshape(-1)


UNET_CFG = FULL_UNET_CFG
DD_CFG = FULL_DD_CFG


# ---- first-stage encoder (row f1): tests/golden/make_encoder_fullsize_golden.py records the REAL reference on these frames
ENC_GOLDEN_FILE = os.path.join(GOLDEN, "encoder_fullsize.npz")
ENC_FRAMES, ENC_H, ENC_W = 16, 320, 512
N_ENC = 65536                      # sampled positions per recorded tensor
ENC_SEEDS = dict(mean=31, logvar=32, hid0=40, hid1=41, hid2=42, hid3=43, hid4=44)


def encoder_frames():
    """16 distinct frames in [-1, 1] (seed only): smooth structure + texture, so every level sees signal."""
    g = torch.Generator().manual_seed(4242)
    base = torch.randn(ENC_FRAMES, 3, ENC_H // 8, ENC_W // 8, generator=g)
    img = torch.nn.functional.interpolate(base, size=(ENC_H, ENC_W), mode="bilinear", align_corners=False)
    img = img + 0.25 * torch.randn(ENC_FRAMES, 3, ENC_H, ENC_W, generator=g)
    return torch.tanh(img)
