"""The reference's YAML (target/params strings quoted here) resolves to the MI355X-native classes once
`dropin.install()` has published them under the reference's dotted paths."""
import importlib
import sys

import yaml

# the structure of configs/inference_512_v1.0.yaml (reference), class paths verbatim
TARGETS = [
    "lvdm.models.ddpm3d.LatentVisualDiffusion",
    "lvdm.modules.networks.openaimodel3d.UNetModel",
    "lvdm.models.autoencoder.AutoencoderKL_Dualref",
    "lvdm.modules.encoders.condition.FrozenOpenCLIPEmbedder",
    "lvdm.modules.encoders.condition.FrozenOpenCLIPImageEmbedderV2",
    "lvdm.modules.encoders.resampler.Resampler",
]


def test_reference_dotted_paths_resolve_to_native_classes():
    from tooncrafter_amd import dropin
    saved = {k: v for k, v in sys.modules.items() if k == "lvdm" or k.startswith("lvdm.") or k in ("utils", "utils.utils")}
    try:
        dropin.install(shims=False)
        for t in TARGETS:
            mod, cls = t.rsplit(".", 1)
            obj = getattr(importlib.import_module(mod), cls)
            assert obj.__module__.startswith("tooncrafter_amd."), (t, obj.__module__)
        from lvdm.models.samplers.ddim import DDIMSampler          # what scripts/evaluation/inference.py:14 does
        from utils.utils import instantiate_from_config              # inference.py:16
        assert DDIMSampler.__module__ == "tooncrafter_amd.lvdm.ddim"
        node = yaml.safe_load("target: lvdm.modules.encoders.resampler.Resampler\nparams: {dim: 1024, num_queries: 16, video_length: 16, output_dim: 1024}")
        assert instantiate_from_config(node).__class__.__name__ == "Resampler"
    finally:
        for k in [k for k in sys.modules if k == "lvdm" or k.startswith("lvdm.") or k in ("utils", "utils.utils")]:
            del sys.modules[k]
        sys.modules.update(saved)
