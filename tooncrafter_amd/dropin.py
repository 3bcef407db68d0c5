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
    if missing("torchvision"):
        _install_torchvision_shim()


def _install_torchvision_shim() -> None:
    """The slice of torchvision the reference's inference script touches (inference.py:9-10,65-69,126-133,
    154-155): PIL-based Resize / CenterCrop / ToTensor / Normalize / Compose, utils.make_grid and io.write_video.
    No encoder library exists on such an image, so write_video goes through tooncrafter_amd.mp4 -- a real H.264 `.mp4`
    under the name the script asked for, every macroblock I_PCM (the same fallback as output.default_writer)."""
    import os

    import numpy as np
    import torch
    from PIL import Image

    tv = types.ModuleType("torchvision")
    tv.__path__ = []
    tv.__tooncrafter_shim__ = True
    tr = types.ModuleType("torchvision.transforms")

    class Compose:
        def __init__(self, transforms):
            self.transforms = list(transforms)

        def __call__(self, x):
            for t in self.transforms:
                x = t(x)
            return x

    class Resize:
        """int size: the SHORTER side becomes `size`, aspect kept (torchvision semantics); bilinear + antialias."""

        def __init__(self, size, interpolation=None, **kw):
            self.size = size

        def __call__(self, img):
            w, h = img.size
            if isinstance(self.size, int):
                if (w <= h and w == self.size) or (h <= w and h == self.size):
                    return img
                if w < h:
                    ow, oh = self.size, int(self.size * h / w)
                else:
                    oh, ow = self.size, int(self.size * w / h)
            else:
                oh, ow = self.size
            return img.resize((ow, oh), Image.BILINEAR)

    class CenterCrop:
        def __init__(self, size):
            self.size = (size, size) if isinstance(size, int) else tuple(size)

        def __call__(self, img):
            w, h = img.size
            th, tw = self.size
            left, top = int(round((w - tw) / 2.0)), int(round((h - th) / 2.0))
            return img.crop((left, top, left + tw, top + th))

    class ToTensor:
        def __call__(self, img):
            a = np.asarray(img, dtype=np.uint8)
            if a.ndim == 2:
                a = a[:, :, None]
            return torch.from_numpy(a.copy()).permute(2, 0, 1).to(torch.float32).div(255.0)

    class Normalize:
        def __init__(self, mean, std, inplace=False):
            self.mean, self.std = torch.tensor(mean).view(-1, 1, 1), torch.tensor(std).view(-1, 1, 1)

        def __call__(self, x):
            return (x - self.mean) / self.std

    tr.Compose, tr.Resize, tr.CenterCrop, tr.ToTensor, tr.Normalize = Compose, Resize, CenterCrop, ToTensor, Normalize
    ut = types.ModuleType("torchvision.utils")

    def make_grid(tensor, nrow=8, padding=2, **kw):
        if padding != 0:
            raise NotImplementedError("shim make_grid: padding=0 only (what the scripts use)")
        t = tensor if tensor.dim() == 4 else tensor.unsqueeze(0)
        rows = [torch.cat(list(t[i:i + nrow]), dim=2) for i in range(0, t.shape[0], nrow)]
        return torch.cat(rows, dim=1)

    ut.make_grid = make_grid
    io = types.ModuleType("torchvision.io")

    def write_video(filename, video_array, fps, video_codec="h264", options=None, **kw):
        from .mp4 import write_mp4
        frames = torch.as_tensor(video_array).cpu()
        os.makedirs(os.path.dirname(filename) or ".", exist_ok=True)
        if frames.shape[1] % 2 or frames.shape[2] % 2:           # yuv420p needs even sizes: keep the raw frames instead
            alt = os.path.splitext(filename)[0] + ".npy"
            np.save(alt, frames.numpy())
            return alt
        return write_mp4(filename, frames, int(fps))

    io.write_video = write_video
    tv.transforms, tv.utils, tv.io = tr, ut, io
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tr, "torchvision.utils": ut, "torchvision.io": io})


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
