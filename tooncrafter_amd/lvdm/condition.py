"""Conditioning stack (the step before the loop -- not on the timed path).

Mirror of reference lvdm/modules/encoders/condition.py: `FrozenOpenCLIPEmbedder` (174-234) and
`FrozenOpenCLIPImageEmbedderV2` (295-372) with the same constructor kwargs and call signatures, on the HIP
kernels (lvdm/openclip.py).  The reference builds the networks with the third-party `open_clip` package, which
this image lacks, so they are rebuilt here from the published ViT-H/14 hyper-parameters with open_clip's
parameter names (towers pinned to HuggingFace transformers' CLIP: tests/test_openclip_golden_cpu.py).  Two host-side pieces of
the reference live in third-party packages that are absent too; both are restated from their published algorithms -- the
tokeniser pinned to transformers.CLIPTokenizer on a shared merge table, the resize pinned to a second, independent statement
of kornia's algorithm on scipy.ndimage + a numpy Keys interpolation (tests/golden/make_resize_golden.py,
tests/test_resize_pin_cpu.py: identical to 1e-12 in float64, 4e-5 as run in fp32):
  * tokenisation (`open_clip.tokenize`, open_clip_torch 2.22.0): `lvdm/clip_tokenizer.py`; the BPE merge table is DATA
    (open_clip's `bpe_simple_vocab_16e6.txt.gz`): with its path in `TC_CLIP_BPE_VOCAB` (or the package installed) any
    prompt is tokenised; without it, already-tokenised int64 (B, 77) tensors and the empty prompt "" (the scripts'
    default, whose tokens do not depend on the table) are accepted and other strings raise;
  * image resize (`kornia.geometry.resize(x, (224, 224), 'bicubic', align_corners=True, antialias=True)`, kornia
    unpinned in requirements.txt): `kornia_resize` below -- kornia's antialias is a Gaussian blur sized from the
    scale factor in front of a plain bicubic interpolation (NOT torch's antialias=True filter).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .openclip import _VisualHolder, build_text


def _gaussian_kernel1d(ks: int, sigma: float, device, dtype) -> torch.Tensor:
    x = torch.arange(ks, device=device, dtype=dtype) - ks // 2
    if ks % 2 == 0:
        x = x + 0.5
    g = torch.exp(-x.pow(2.0) / (2.0 * sigma ** 2))
    return g / g.sum()


def kornia_resize(x: torch.Tensor, size, interpolation: str = "bicubic", align_corners: bool = True,
                  antialias: bool = True) -> torch.Tensor:
    """Restatement of `kornia.geometry.transform.resize` (kornia/geometry/transform/affwarp.py) for (B, C, H, W) input,
    as `FrozenOpenCLIPImageEmbedderV2.preprocess` calls it (reference condition.py:322-326).  When downscaling with
    antialias, kornia first blurs with a separable Gaussian -- sigma = (scale factor - 1) / 2 per axis (at least 0.001),
    kernel size int(max(4 sigma, 3)) made odd, reflect border (kornia/filters/gaussian.py: gaussian_blur2d) -- and then
    calls torch.nn.functional.interpolate WITHOUT torch's own antialias flag."""
    h, w = x.shape[-2:]
    factors = (h / size[0], w / size[1])
    if antialias and max(factors) > 1:
        sig = [max((f - 1.0) / 2.0, 0.001) for f in factors]
        ks = [int(max(2.0 * 2 * s, 3)) for s in sig]
        ks = [k + 1 if k % 2 == 0 else k for k in ks]
        c = x.shape[1]
        ky = _gaussian_kernel1d(ks[0], sig[0], x.device, x.dtype).view(1, 1, ks[0], 1).expand(c, 1, ks[0], 1)
        kx = _gaussian_kernel1d(ks[1], sig[1], x.device, x.dtype).view(1, 1, 1, ks[1]).expand(c, 1, 1, ks[1])
        x = F.pad(x, (ks[1] // 2, ks[1] // 2, ks[0] // 2, ks[0] // 2), mode="reflect")
        x = F.conv2d(F.conv2d(x, kx, groups=c), ky, groups=c)
    return F.interpolate(x, size=tuple(size), mode=interpolation, align_corners=align_corners)


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
        if getattr(self, "_bpe", None) is None:
            from .clip_tokenizer import CLIPTokenizer, default_vocab_path
            vocab = default_vocab_path()
            if vocab is None:
                raise RuntimeError("tokenising strings needs CLIP's BPE merge table: set TC_CLIP_BPE_VOCAB to open_clip's "
                                   "bpe_simple_vocab_16e6.txt.gz (or install open_clip), or pass an int64 (B, 77) token tensor")
            self._bpe = CLIPTokenizer(vocab, context_length=self.max_length)
        return self._bpe(text)

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
        x = kornia_resize(x.float(), (size, size), interpolation="bicubic", align_corners=True, antialias=self.antialias)
        x = (x + 1.) / 2.
        return (x - self.mean.view(1, 3, 1, 1)) / self.std.view(1, 3, 1, 1)

    def forward(self, image, no_dropout=False):
        return self.encode_with_vision_transformer(image)

    def encode_with_vision_transformer(self, x):
        return self.model.visual.tokens(self.preprocess(x))
