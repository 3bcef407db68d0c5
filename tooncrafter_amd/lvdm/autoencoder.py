"""First-stage model wrapper.  Interface of reference lvdm/models/autoencoder.py:
AutoencoderKL.decode (112-116) and AutoencoderKL_Dualref (238-275).

Only the decode side is on the hot path `north_star` names.  The encoder
(lvdm/modules/networks/ae_modules.py:366-475) runs once per clip
before the loop and is the next row to build; until then `encode` raises instead of
falling back to anything.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .autoencoder_dualref import VideoDecoder


class AutoencoderKL_Dualref(nn.Module):
    def __init__(self, ddconfig, lossconfig=None, embed_dim=4, ckpt_path=None, ignore_keys=[], image_key="image",
                 colorize_nlabels=None, monitor=None, test=False, logdir=None, input_dim=4, test_args=None,
                 additional_decode_keys=None, use_checkpoint=False, diff_boost_factor=3.0):
        super().__init__()
        dd = dict(ddconfig)
        assert dd["double_z"]
        self.embed_dim = embed_dim
        self.image_key = image_key
        self.decoder = VideoDecoder(**dd)
        # 1x1 convs of the KL autoencoder; post_quant_conv is bypassed by the video decode path
        # (autoencoder.py:113-114) but is part of the checkpoint.
        self.quant_conv = nn.Conv2d(2 * dd["z_channels"], 2 * embed_dim, 1)
        self.post_quant_conv = nn.Conv2d(embed_dim, dd["z_channels"], 1)
        if monitor is not None:
            self.monitor = monitor

    @property
    def device(self):
        return next(self.parameters()).device

    def encode(self, x, return_hidden_states=False, **kwargs):
        raise NotImplementedError(
            "the first-stage Encoder is SURVEY.md row f1 (next to build); the hot path takes the "
            "latent and the five reference hidden states as inputs")

    def decode(self, z, **kwargs):
        """z: (B*T, zc, h, w) already divided by scale_factor.  With kwargs (ref_context,
        timesteps) the reference skips post_quant_conv and calls the video decoder."""
        if len(kwargs) == 0:
            raise NotImplementedError("image-only decode (post_quant_conv + plain decoder) is off the hot path")
        return self.decoder(z, **kwargs)
