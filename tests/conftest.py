import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


def load_golden(name):
    return {k: v for k, v in np.load(os.path.join(GOLDEN, name)).items()}


def sub_state_dict(sd, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


@pytest.fixture(scope="session")
def tiny_sd(manifest):
    from tooncrafter_amd.synth import synth_state_dict
    return synth_state_dict({k: tuple(v) for k, v in manifest["tiny"].items()}, seed=1234)


TINY_UNET_CFG = dict(in_channels=8, out_channels=4, model_channels=64, num_res_blocks=2,
                     attention_resolutions=[4, 2, 1], channel_mult=[1, 2, 4, 4], num_head_channels=64,
                     transformer_depth=1, context_dim=96, use_linear=True, temporal_conv=True,
                     temporal_attention=True, temporal_selfatt_only=True, use_relative_position=False,
                     use_causal_attention=False, temporal_length=4, addition_attention=True,
                     image_cross_attention=True, default_fs=24, fs_condition=True, dropout=0.1,
                     use_checkpoint=False)
TINY_DD_CFG = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=64,
                   ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)
FULL_UNET_CFG = dict(TINY_UNET_CFG, model_channels=320, context_dim=1024, temporal_length=16)
FULL_DD_CFG = dict(TINY_DD_CFG, ch=128)


def rel_l2(a, b):
    a = a.double().flatten()
    b = b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))
