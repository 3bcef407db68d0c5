"""Synthetic stand-ins for the conditioning stack (NOT on the hot path).

The reference's `FrozenOpenCLIPEmbedder` and `FrozenOpenCLIPImageEmbedderV2`
(lvdm/modules/encoders/condition.py:174-234, 295-372) need the third-party open_clip
package + downloaded ViT-H/14 weights, neither of which exists on the build or GPU
boxes.  (The `Resampler` that follows them is real: lvdm/resampler.py.)  So that
`configs/inference_512_v1.0.yaml` instantiates unmodified, these classes accept the
same constructor kwargs and produce tensors of the right shape from a seeded
generator.  They carry no parameters and do no real conditioning; a deployment
supplies the real embedders (any object with the same call signature).
"""
from __future__ import annotations

import zlib

import torch
import torch.nn as nn


def _seeded(shape, key: str, device):
    g = torch.Generator().manual_seed(zlib.crc32(key.encode()) & 0x7FFFFFFF)
    return torch.randn(shape, generator=g).to(device)


class _Stub(nn.Module):
    def __init__(self, **kwargs):
        super().__init__()
        self.kwargs = kwargs
        self._dev = nn.Parameter(torch.zeros(()), requires_grad=False)


class FrozenOpenCLIPEmbedder(_Stub):
    """text -> (B, 77, 1024)"""

    def forward(self, text):
        return torch.cat([_seeded((1, 77, 1024), str(t), self._dev.device) for t in text], 0)

    encode = forward


class FrozenOpenCLIPImageEmbedderV2(_Stub):
    """image (B, 3, H, W) -> (B, 257, 1280)"""

    def forward(self, image):
        return torch.cat([_seeded((1, 257, 1280), f"{float(im.float().mean()):.6f}", image.device) for im in image], 0)
