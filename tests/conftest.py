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


@pytest.fixture(scope="session", autouse=True)
def gpu_preflight(request):
    """Before the first GPU test: a plain-PyTorch health check of the box in a SUBPROCESS (scripts/gpu_health.py:
    copies, elementwise, GEMM, 64 GiB fill, hipGraph replay -- libtooncrafter_hip.so is not loaded there).
    One GPU of the pool faults on any sustained work (round-1 driver run, several round-2 leases: always
    'Memory access fault by GPU node-2', see profiles/r02_node2_*): when the box itself is unhealthy say so and
    stop, instead of letting the fault be attributed to whichever kernel happened to run."""
    if not torch.cuda.is_available() or not any("gpu" in item.keywords for item in request.session.items):
        return
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gpu_health.py")], capture_output=True,
                           text=True, timeout=600)
        ok, tail = r.returncode == 0, (r.stdout + r.stderr)[-800:]
    except subprocess.TimeoutExpired:
        ok, tail = False, "gpu_health.py timed out after 600 s"
    if not ok:
        pytest.exit("GPU BOX UNHEALTHY: scripts/gpu_health.py (plain PyTorch, no tooncrafter code loaded) failed on this "
                    "lease -- the failure below is the box's, not the kernels':\n" + tail, returncode=3)


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


# ---------------------------------------------------------------- full-size fixtures (BASELINE.json configs[1] shapes)
# Session-scoped: the 1.44 B-parameter model is synthesised once for tests/test_gpu_fullsize.py and
# tests/test_gpu_ddim50.py.
@pytest.fixture(scope="session")
def golden():
    import fullsize_cases as fc
    if not os.path.exists(fc.GOLDEN_FILE):
        pytest.fail(f"{fc.GOLDEN_FILE} missing: run tests/golden/make_fullsize_golden.py")
    return dict(np.load(fc.GOLDEN_FILE))


@pytest.fixture(scope="session")
def inp():
    import fullsize_cases as fc
    return fc.inputs()


@pytest.fixture(scope="session")
def full_model():
    """LatentVisualDiffusion at the full widths on the HIP backend, parameters drawn on the CPU generator (the values the
    oracle goldens were made with), one tensor at a time."""
    import bench
    from tooncrafter_amd import ops, synth
    from tooncrafter_amd.utils import instantiate_from_config
    assert ops.backend().name == "hip"
    with torch.device("meta"):
        model = instantiate_from_config(dict(target="lvdm.models.ddpm3d.LatentVisualDiffusion",
                                             params=bench.MODEL_PARAMS))
    model = model.to_empty(device="cuda").eval()
    with torch.no_grad():
        for name, p in model.named_parameters():
            p.copy_(synth.synth_tensor(name, tuple(p.shape), 1234, "cpu"))
        bufs = bench.instantiate_schedule()
        for name, b in model.named_buffers():
            b.copy_(bufs[name].to("cuda"))
    return model
