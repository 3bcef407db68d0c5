"""Publish the MI355X-native mirror under the reference's dotted module paths.

`configs/inference_512_v1.0.yaml` names classes such as
`lvdm.models.ddpm3d.LatentVisualDiffusion`, and the reference scripts import
`lvdm.models.samplers.ddim.DDIMSampler` and `utils.utils.instantiate_from_config`
(scripts/evaluation/inference.py:14-16).  `install()` registers this package's modules in
`sys.modules` under exactly those names, so both resolve to the HIP path without touching
the YAML or the scripts:

    python -m tooncrafter_amd.dropin /path/to/ToonCrafter/scripts/evaluation/inference.py --config ... 

Packages the reference scripts import but this image lacks (omegaconf, pytorch_lightning,
torchvision) get minimal shims only if they are genuinely absent.
"""
from __future__ import annotations

import importlib
import runpy
import sys
import types

from .utils import MODULE_ALIASES


def _pkg(name: str) -> types.ModuleType:
    m = sys.modules.get(name)
    if m is None:
        m = types.ModuleType(name)
        m.__path__ = []          # mark as package
        sys.modules[name] = m
    return m


def install(shims: bool = True) -> None:
    for ref_name, ours in MODULE_ALIASES.items():
        mod = importlib.import_module(ours)
        parts = ref_name.split(".")
        for i in range(1, len(parts)):
            parent = _pkg(".".join(parts[:i]))
            child_name = ".".join(parts[:i + 1])
            child = mod if i == len(parts) - 1 else _pkg(child_name)
            sys.modules[child_name] = child
            setattr(parent, parts[i], child)
    if shims:
        _install_shims()


def _install_shims() -> None:
    import torch
    import yaml

    def missing(name):
        try:
            importlib.import_module(name)
            return False
        except Exception:
            return True

    if missing("omegaconf"):
        class DictConfig(dict):
            def __getattr__(self, k):
                try:
                    return self[k]
                except KeyError as e:
                    raise AttributeError(k) from e

            def __setattr__(self, k, v):
                self[k] = v

            def pop(self, k, *d):
                return dict.pop(self, k, *d)

        def wrap(o):
            if isinstance(o, dict):
                return DictConfig({k: wrap(v) for k, v in o.items()})
            if isinstance(o, list):
                return [wrap(v) for v in o]
            return o

        class OmegaConf:
            @staticmethod
            def load(path):
                with open(path) as f:
                    return wrap(yaml.safe_load(f))

            @staticmethod
            def create(o=None):
                return wrap(o or {})

        m = types.ModuleType("omegaconf")
        m.OmegaConf, m.DictConfig = OmegaConf, DictConfig
        sys.modules["omegaconf"] = m
    if missing("pytorch_lightning"):
        m = types.ModuleType("pytorch_lightning")

        def seed_everything(seed):
            import random
            import numpy as np
            random.seed(seed)
            np.random.seed(seed)
            torch.manual_seed(seed)
            return seed
        m.seed_everything = seed_everything
        m.LightningModule = torch.nn.Module
        sys.modules["pytorch_lightning"] = m


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        print(__doc__)
        return 2
    install()
    sys.argv = argv
    runpy.run_path(argv[0], run_name="__main__")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
