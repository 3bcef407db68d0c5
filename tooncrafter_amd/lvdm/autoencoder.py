"""First-stage model wrapper.  Interface of reference lvdm/models/autoencoder.py:
AutoencoderKL.encode (100-110) / .decode (112-116) and AutoencoderKL_Dualref (238-275), plus
the posterior object of lvdm/distributions.py:24-65.

decode is on the hot path `north_star` names; encode runs once per clip
before the loop and shares the decoder's kernels.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .ae_modules import Encoder
from .autoencoder_dualref import VideoDecoder
from .common import f32, pack_conv3x3


class DiagonalGaussianDistribution(object):
    """Posterior q(z|x) = N(mean, exp(logvar)); `parameters` = (N, 2z, h, w) moments."""

    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)
        if deterministic:
            self.var = self.std = torch.zeros_like(self.mean)

    def sample(self, noise=None):
        if noise is None:
            noise = torch.randn(self.mean.shape)          # the reference draws this on the CPU (distributions.py:37)
        return self.mean + self.std * noise.to(device=self.parameters.device)

    def mode(self):
        return self.mean


class AutoencoderKL_Dualref(nn.Module):
    def __init__(self, ddconfig, lossconfig=None, embed_dim=4, ckpt_path=None, ignore_keys=[], image_key="image",
                 colorize_nlabels=None, monitor=None, test=False, logdir=None, input_dim=4, test_args=None,
                 additional_decode_keys=None, use_checkpoint=False, diff_boost_factor=3.0):
        super().__init__()
        dd = dict(ddconfig)
        assert dd["double_z"]
        self.embed_dim = embed_dim
        self.image_key = image_key
        self.encoder = Encoder(**dd)
        self.decoder = VideoDecoder(**dd)
        # 1x1 convs of the KL autoencoder; post_quant_conv is bypassed by the video decode path
        # (autoencoder.py:113-114) but is part of the checkpoint.
        self.quant_conv = nn.Conv2d(2 * dd["z_channels"], 2 * embed_dim, 1)
        self.post_quant_conv = nn.Conv2d(embed_dim, dd["z_channels"], 1)
        if monitor is not None:
            self.monitor = monitor
        self._quant_pack = None

    @property
    def device(self):
        return next(self.parameters()).device

    def _apply(self, fn, recurse=True):
        self._quant_pack = None
        return super()._apply(fn, recurse)

    def _fused_out_conv(self):
        """quant_conv (1x1) composed into the encoder's last 3x3 conv:
        Wq (Wo * x + bo) + bq = (Wq Wo) * x + (Wq bo + bq), composed in fp32 -- one GEMM, and the
        moments never pass through a bf16 intermediate."""
        key = (self.quant_conv.weight._version, self.encoder.conv_out.weight._version, self.quant_conv.weight.device)
        if self._quant_pack is None or self._quant_pack[0] != key:
            with torch.no_grad():
                wq = self.quant_conv.weight.float()[:, :, 0, 0]
                wo = self.encoder.conv_out.weight.float()
                w = torch.einsum("oc,cikl->oikl", wq, wo)
                b = wq @ self.encoder.conv_out.bias.float() + self.quant_conv.bias.float()
                self._quant_pack = (key, pack_conv3x3(w), f32(b))
        return self._quant_pack[1], self._quant_pack[2]

    def encode(self, x, return_hidden_states=False, **kwargs):
        """x: (N, 3, H, W) in [-1, 1] -> posterior [, hidden states as (N, C, H_l, W_l) fp32 tensors]."""
        from .. import ops
        enc = self.encoder
        w, b = self._fused_out_conv()
        moments, hidden, h, wd = enc.encode_rows(x, out_w=w, out_b=b)
        n = x.shape[0]
        mom = ops.rows_to_nchw(moments, c=moments.shape[1], b=n, t=1, h=h, w=wd)[:, :, 0]
        posterior = DiagonalGaussianDistribution(mom)
        if not return_hidden_states:
            return posterior
        hs = [ops.rows_to_nchw(a.rows, c=a.c, b=n, t=1, h=a.h, w=a.w)[:, :, 0] for a in hidden]
        return posterior, hs

    def decode(self, z, **kwargs):
        """z: (B*T, zc, h, w) already divided by scale_factor.  With kwargs (ref_context,
        timesteps) the reference skips post_quant_conv and calls the video decoder."""
        if len(kwargs) == 0:
            raise NotImplementedError("image-only decode (post_quant_conv + plain decoder) is off the hot path")
        return self.decoder(z, **kwargs)
