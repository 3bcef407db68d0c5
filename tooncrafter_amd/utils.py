"""The reference's plugin mechanism (utils/utils.py:27-42): a YAML node
`{target: dotted.Class, params: {...}}` is instantiated by importing the class.
Targets under the reference's `lvdm.*` namespace resolve to this package's
MI355X-native mirror of the same class."""
from __future__ import annotations

import importlib

# reference dotted module path -> module of this package implementing the same interface
MODULE_ALIASES = {
    "lvdm.models.ddpm3d": "tooncrafter_amd.lvdm.ddpm3d",
    "lvdm.models.autoencoder": "tooncrafter_amd.lvdm.autoencoder",
    "lvdm.models.autoencoder_dualref": "tooncrafter_amd.lvdm.autoencoder_dualref",
    "lvdm.models.utils_diffusion": "tooncrafter_amd.lvdm.utils_diffusion",
    "lvdm.models.samplers.ddim": "tooncrafter_amd.lvdm.ddim",
    "lvdm.models.samplers.ddim_multiplecond": "tooncrafter_amd.lvdm.ddim_multiplecond",
    "lvdm.modules.attention": "tooncrafter_amd.lvdm.attention",
    "lvdm.modules.networks.openaimodel3d": "tooncrafter_amd.lvdm.openaimodel3d",
    "lvdm.modules.encoders.condition": "tooncrafter_amd.lvdm.condition",
    "lvdm.modules.encoders.resampler": "tooncrafter_amd.lvdm.resampler",
    "utils.utils": "tooncrafter_amd.utils",
}


def get_obj_from_str(string, reload=False):
    module, cls = string.rsplit(".", 1)
    module = MODULE_ALIASES.get(module, module)
    mod = importlib.import_module(module)
    if reload:
        importlib.reload(mod)
    return getattr(mod, cls)


def _get(node, key, default=None):
    if isinstance(node, dict):
        return node.get(key, default)
    return getattr(node, key, default) if hasattr(node, key) else default


def instantiate_from_config(config):
    target = _get(config, "target")
    if target is None:
        if config in ('__is_first_stage__', '__is_unconditional__'):
            return None
        raise KeyError("Expected key `target` to instantiate.")
    params = _get(config, "params", None) or {}
    return get_obj_from_str(target)(**dict(params))


def count_params(model, verbose=False):
    n = sum(p.numel() for p in model.parameters())
    if verbose:
        print(f"{model.__class__.__name__} has {n * 1.e-6:.2f} M params.")
    return n
