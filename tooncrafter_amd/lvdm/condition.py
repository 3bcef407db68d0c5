"""Conditioning stack (the step before the loop -- not on the timed path).

Mirror of reference lvdm/modules/encoders/condition.py: `FrozenOpenCLIPEmbedder` (174-234) and
`FrozenOpenCLIPImageEmbedderV2` (295-372) with the same constructor kwargs and call signatures, on the HIP
kernels (lvdm/openclip.py).  The reference builds the networks with the third-party `open_clip` package, which
this image lacks, so they are rebuilt here from the published ViT-H/14 hyper-parameters with open_clip's
parameter names; two host-side pieces of the reference cannot be reproduced exactly without their packages and
say so loudly instead of guessing:
  * tokenisation (`open_clip.tokenize`, BPE vocabulary file): `forward(text)` accepts already-tokenised
    int64 (B, 77) tensors and the empty prompt "" (the scripts' default); other strings need `open_clip`;
  * image resize (`kornia.geometry.resize(..., 'bicubic', align_corners=True, antialias=True)`): replaced by
    `torch.nn.functional.interpolate(bicubic, align_corners=True, antialias=True)` -- a different antialias
    filter, i.e. a documented deviation of the preprocessing, not of the towers.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .openclip import _VisualHolder, build_text


class AbstractEncoder(nn.Module):
    def encode(self, *args, **kwargs):
        raise NotImplementedError


class FrozenOpenCLIPEmbedder(AbstractEncoder):
    """text -> (B, 77, 1024): OpenCLIP text transformer, penultimate layer, ln_final (no pooling)."""
    LAYERS = ["last", "penultimate"]

    def __init__(self, arch="ViT-H-14", version="laion2b_s32b_b79k", device="cuda", max_length=77, freeze=True,
                 layer="last"):
        super().__init__()
        assert layer in self.LAYERS
        self.model = build_text(arch)
        self.device, self.max_length, self.layer = device, max_length, layer
        self.layer_idx = 0 if layer == "last" else 1
        if freeze:
            self.freeze()

    def freeze(self):
        self.model = self.model.eval()
        for p in self.parameters():
            p.requires_grad = False

    def tokenize(self, text):
        text = [text] if isinstance(text, str) else list(text)
        if all(t == "" for t in text):
            # the only prompts the interpolation scripts use by default (inference.py:186-187,209-210):
            # "" tokenises to <start_of_text>=49406, <end_of_text>=49407, zero padding (open_clip/tokenizer.py)
            tok = torch.zeros((len(text), self.max_length), dtype=torch.long)
            tok[:, 0], tok[:, 1] = 49406, 49407
            return tok
        try:
            import open_clip
        except Exception as e:
            raise RuntimeError("tokenising strings needs the `open_clip` package (BPE vocabulary); pass an int64 "
                               "(B, 77) token tensor instead") from e
        return open_clip.tokenize(text)

    def forward(self, text):
        tokens = text if torch.is_tensor(text) else self.tokenize(text)
        dev = self.model.positional_embedding.device
        return self.encode_with_transformer(tokens.to(dev))

    def encode_with_transformer(self, text):
        return self.model.tokens(text, skip_last=self.layer_idx)

    def encode(self, text):
        return self(text)


class FrozenOpenCLIPImageEmbedderV2(AbstractEncoder):
    """image (B, 3, H, W) in [-1, 1] -> (B, 257, 1280): all tokens of the OpenCLIP vision transformer."""

    def __init__(self, arch="ViT-H-14", version="laion2b_s32b_b79k", device="cuda", freeze=True, layer="pooled",
                 antialias=True):
        super().__init__()
        self.model = _VisualHolder(arch)
        self.device, self.layer, self.antialias = device, layer, antialias
        if layer == "penultimate":
            raise NotImplementedError()
        if freeze:
            self.freeze()
        self.register_buffer('mean', torch.Tensor([0.48145466, 0.4578275, 0.40821073]), persistent=False)
        self.register_buffer('std', torch.Tensor([0.26862954, 0.26130258, 0.27577711]), persistent=False)

    def freeze(self):
        self.model = self.model.eval()
        for p in self.model.parameters():
            p.requires_grad = False

    def preprocess(self, x):
        size = self.model.visual.grid_size[0] * self.model.visual.patch_size[0]
        x = F.interpolate(x.float(), size=(size, size), mode="bicubic", align_corners=True, antialias=self.antialias)
        x = (x + 1.) / 2.
        return (x - self.mean.view(1, 3, 1, 1)) / self.std.view(1, 3, 1, 1)

    def forward(self, image, no_dropout=False):
        return self.encode_with_vision_transformer(image)

    def encode_with_vision_transformer(self, x):
        return self.model.visual.tokens(self.preprocess(x))
