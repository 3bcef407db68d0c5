"""Pipeline model: schedule buffers, apply_model, v-parameterisation algebra,
decode_first_stage.  Interface of reference lvdm/models/ddpm3d.py -- DDPM (41-463:
register_schedule 124-187, predict_* 240-252), LatentDiffusion (465-1039: scale_arr
523-528, decode_core 647-679, apply_model 735-750), LatentVisualDiffusion
(1041-1240), DiffusionWrapper (1243-1310, `hybrid` branch) -- restricted to what the
inference scripts touch.  Training, losses, logging and EMA are out of scope.
"""
from __future__ import annotations

import os
from functools import partial

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from ..utils import instantiate_from_config
from .common import PackedModule, SourceKey
from .utils_diffusion import make_beta_schedule, rescale_zero_terminal_snr


def _get(node, key, default=None):
    if isinstance(node, dict):
        return node.get(key, default)
    return getattr(node, key, default)


class _DeviceModule(nn.Module):
    """Stands in for pl.LightningModule: an nn.Module with a `.device` property."""

    @property
    def device(self):
        for p in self.parameters():
            return p.device
        for b in self.buffers():
            return b.device
        return torch.device("cpu")


class DiffusionWrapper(_DeviceModule):
    def __init__(self, diff_model_config, conditioning_key):
        super().__init__()
        self.diffusion_model = instantiate_from_config(diff_model_config)
        self.conditioning_key = conditioning_key

    def forward(self, x, t, c_concat: list = None, c_crossattn: list = None, c_adm=None, s=None, mask=None,
                **kwargs):
        if self.conditioning_key == 'hybrid':
            # x (+) c_concat on the channel axis is fused into the UNet's input layout converter
            cc = c_crossattn[0] if len(c_crossattn) == 1 else torch.cat(c_crossattn, 1)
            parts = [x] + list(c_concat)
            if len(parts) > 2:
                parts = [x, torch.cat(list(c_concat), dim=1)]
            return self.diffusion_model(None, t, context=cc, x_parts=parts, **kwargs)
        if self.conditioning_key == 'crossattn':
            cc = c_crossattn[0] if len(c_crossattn) == 1 else torch.cat(c_crossattn, 1)
            return self.diffusion_model(x, t, context=cc, **kwargs)
        raise NotImplementedError(f"conditioning_key '{self.conditioning_key}' (the config uses 'hybrid')")


class DDPM(_DeviceModule):
    def __init__(self, unet_config, timesteps=1000, beta_schedule="linear", loss_type="l2", ckpt_path=None,
                 ignore_keys=[], load_only_unet=False, monitor=None, use_ema=True, first_stage_key="image",
                 image_size=256, channels=3, log_every_t=100, clip_denoised=True, linear_start=1e-4,
                 linear_end=2e-2, cosine_s=8e-3, given_betas=None, original_elbo_weight=0., v_posterior=0.,
                 l_simple_weight=1., conditioning_key=None, parameterization="eps", scheduler_config=None,
                 use_positional_encodings=False, learn_logvar=False, logvar_init=0.,
                 rescale_betas_zero_snr=False):
        super().__init__()
        assert parameterization in ["eps", "x0", "v"]
        if use_ema:
            raise NotImplementedError("EMA is a training feature (use_ema: False in the inference config)")
        self.parameterization = parameterization
        self.cond_stage_model = None
        self.clip_denoised = clip_denoised
        self.log_every_t = log_every_t
        self.first_stage_key = first_stage_key
        self.channels = channels
        self.temporal_length = _get(_get(unet_config, "params"), "temporal_length")
        self.image_size = [image_size, image_size] if isinstance(image_size, int) else image_size
        self.model = DiffusionWrapper(unet_config, conditioning_key)
        self.use_ema = use_ema
        self.rescale_betas_zero_snr = rescale_betas_zero_snr
        self.v_posterior = v_posterior
        if monitor is not None:
            self.monitor = monitor
        self.register_schedule(given_betas=given_betas, beta_schedule=beta_schedule, timesteps=timesteps,
                               linear_start=linear_start, linear_end=linear_end, cosine_s=cosine_s)

    def register_schedule(self, given_betas=None, beta_schedule="linear", timesteps=1000, linear_start=1e-4,
                          linear_end=2e-2, cosine_s=8e-3):
        betas = given_betas if given_betas is not None else make_beta_schedule(
            beta_schedule, timesteps, linear_start=linear_start, linear_end=linear_end, cosine_s=cosine_s)
        if self.rescale_betas_zero_snr:
            betas = rescale_zero_terminal_snr(betas)
        alphas = 1. - betas
        ac = np.cumprod(alphas, axis=0)
        ac_prev = np.append(1., ac[:-1])
        self.num_timesteps = int(betas.shape[0])
        self.linear_start, self.linear_end = linear_start, linear_end
        t32 = partial(torch.tensor, dtype=torch.float32)
        reg = self.register_buffer
        reg('betas', t32(betas))
        reg('alphas_cumprod', t32(ac))
        reg('alphas_cumprod_prev', t32(ac_prev))
        reg('sqrt_alphas_cumprod', t32(np.sqrt(ac)))
        reg('sqrt_one_minus_alphas_cumprod', t32(np.sqrt(1. - ac)))
        reg('log_one_minus_alphas_cumprod', t32(np.log(1. - ac)))
        if self.parameterization != 'v':
            reg('sqrt_recip_alphas_cumprod', t32(np.sqrt(1. / ac)))
            reg('sqrt_recipm1_alphas_cumprod', t32(np.sqrt(1. / ac - 1)))
        else:
            reg('sqrt_recip_alphas_cumprod', torch.zeros(self.num_timesteps))
            reg('sqrt_recipm1_alphas_cumprod', torch.zeros(self.num_timesteps))
        post_var = (1 - self.v_posterior) * betas * (1. - ac_prev) / (1. - ac) + self.v_posterior * betas
        reg('posterior_variance', t32(post_var))
        reg('posterior_log_variance_clipped', t32(np.log(np.maximum(post_var, 1e-20))))
        reg('posterior_mean_coef1', t32(betas * np.sqrt(ac_prev) / (1. - ac)))
        reg('posterior_mean_coef2', t32((1. - ac_prev) * np.sqrt(alphas) / (1. - ac)))

    # v-parameterisation algebra on torch tensors (API parity; the sampler uses the fused kernel)
    def _at(self, buf, t, x):
        return buf.gather(-1, t).reshape(t.shape[0], *((1,) * (x.dim() - 1)))

    def predict_start_from_z_and_v(self, x_t, t, v):
        return self._at(self.sqrt_alphas_cumprod, t, x_t) * x_t - self._at(self.sqrt_one_minus_alphas_cumprod, t, x_t) * v

    def predict_eps_from_z_and_v(self, x_t, t, v):
        return self._at(self.sqrt_alphas_cumprod, t, x_t) * v + self._at(self.sqrt_one_minus_alphas_cumprod, t, x_t) * x_t


class LatentDiffusion(DDPM):
    def __init__(self, first_stage_config, cond_stage_config, num_timesteps_cond=None, cond_stage_key="caption",
                 cond_stage_trainable=False, cond_stage_forward=None, conditioning_key=None, uncond_prob=0.2,
                 uncond_type="empty_seq", scale_factor=1.0, scale_by_std=False, encoder_type="2d",
                 only_model=False, noise_strength=0, use_dynamic_rescale=False, base_scale=0.7, turning_step=400,
                 loop_video=False, fps_condition_type='fs', perframe_ae=False, logdir=None, rand_cond_frame=False,
                 en_and_decode_n_samples_a_time=None, *args, **kwargs):
        self.num_timesteps_cond = 1 if num_timesteps_cond is None else num_timesteps_cond
        self.scale_by_std = scale_by_std
        kwargs.pop("ckpt_path", None)
        kwargs.pop("ignore_keys", None)
        conditioning_key = 'crossattn' if conditioning_key is None else conditioning_key
        super().__init__(conditioning_key=conditioning_key, *args, **kwargs)
        self.cond_stage_trainable = cond_stage_trainable
        self.cond_stage_key = cond_stage_key
        self.noise_strength = noise_strength
        self.use_dynamic_rescale = use_dynamic_rescale
        self.loop_video = loop_video
        self.fps_condition_type = fps_condition_type
        self.perframe_ae = perframe_ae
        self.en_and_decode_n_samples_a_time = en_and_decode_n_samples_a_time
        if scale_by_std:
            self.register_buffer('scale_factor', torch.tensor(scale_factor))
        else:
            self.scale_factor = scale_factor
        if use_dynamic_rescale:
            arr = np.concatenate((np.linspace(1.0, base_scale, turning_step), np.full(self.num_timesteps, base_scale)))
            self.register_buffer('scale_arr', torch.tensor(arr, dtype=torch.float32))
        self.first_stage_model = instantiate_from_config(first_stage_config).eval()
        for p in self.first_stage_model.parameters():
            p.requires_grad = False
        self.cond_stage_model = self._instantiate_cond_stage(cond_stage_config)
        self.first_stage_config, self.cond_stage_config = first_stage_config, cond_stage_config
        self.clip_denoised = False
        self.encoder_type = encoder_type
        self.uncond_prob, self.uncond_type = uncond_prob, uncond_type
        self.classifier_free_guidance = uncond_prob > 0
        # batched-CFG state: static 2B inputs (+ the captured hipGraph of the UNet forward)
        self._cfg_state = None
        self.use_hipgraph = os.environ.get("TC_HIPGRAPH", "1") != "0"
        self.cfg_share = os.environ.get("TC_CFG_SHARE", "1") != "0"
        # the guided passes behind the shared prefix as concurrent batch-b walks on their own HIP streams
        # (openaimodel3d.UNetModel._forward_branches) instead of one batch-(n b) walk
        self.cfg_streams = os.environ.get("TC_CFG_STREAMS", "0") == "1"

    def _instantiate_cond_stage(self, config):
        model = instantiate_from_config(config)
        if isinstance(model, nn.Module):
            model = model.eval()
            for p in model.parameters():
                p.requires_grad = False
        return model

    def get_learned_conditioning(self, c):
        enc = getattr(self.cond_stage_model, "encode", None)
        return enc(c) if callable(enc) else self.cond_stage_model(c)

    def get_first_stage_encoding(self, encoder_posterior, noise=None):
        z = encoder_posterior.sample(noise=noise) if hasattr(encoder_posterior, "sample") else encoder_posterior
        return self.scale_factor * z

    @torch.no_grad()
    def encode_first_stage(self, x):
        """ddpm3d.py:620-644: (B, 3, T, H, W) or (N, 3, H, W) pixels -> scaled latent.  All frames go
        through the encoder in one call (the reference's perframe_ae loop computes the same thing one
        frame at a time to save memory; 288 GB make that unnecessary)."""
        if x.dim() == 5:
            b, c, t, h, w = x.shape
            frames = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
            z = self.get_first_stage_encoding(self.first_stage_model.encode(frames)).detach()
            return z.reshape(b, t, *z.shape[1:]).permute(0, 2, 1, 3, 4)
        return self.get_first_stage_encoding(self.first_stage_model.encode(x)).detach()

    # ------------------------------------------------------------------ hot path
    def apply_model(self, x_noisy, t, cond, **kwargs):
        if not isinstance(cond, dict):
            cond = {'c_concat' if self.model.conditioning_key == 'concat' else 'c_crossattn':
                    cond if isinstance(cond, list) else [cond]}
        out = self.model(x_noisy, t, **cond, **kwargs)
        return out[0] if isinstance(out, tuple) else out

    def apply_model_cfg(self, x_noisy, t, cond, uncond, **kwargs):
        """Conditional and unconditional passes as ONE batch-2B UNet call.  Exact: every
        normalisation and attention in the UNet is per sample."""
        return tuple(self.apply_model_multi(x_noisy, t, [cond, uncond], **kwargs))

    def apply_model_multi(self, x_noisy, t, conds, **kwargs):
        """The UNet passes of one guided step -- 2 for ddim.py:226-233, 3 for
        ddim_multiplecond.py:226-236 -- as ONE batch-(n B) call; returns one output per entry of
        `conds`, in order."""
        n = len(conds)
        if self.model.conditioning_key != 'hybrid':
            return [self.apply_model(x_noisy, t, c, **kwargs) for c in conds]
        b = x_noisy.shape[0]
        fs = kwargs.get("fs")
        # identity of the conditioning: the tensor OBJECTS (strong references held by SourceKey) and their
        # versions -- never data_ptr(), which the next clip's freshly allocated tensors can share
        cond_t = [tns for c in conds for tns in (*c["c_crossattn"], *c["c_concat"])] + [fs]
        # every pass has the same latent, timestep and fps; when they also share the concat conditioning (the SAME tensor
        # objects: inference.py:213-214 builds `uc` from the very `img_cat_cond` of `cond`) they differ only through the
        # cross-attention context and the UNet computes what precedes it once (common.CfgShare; TC_CFG_SHARE=0: off)
        share = n > 1 and self.cfg_share and all(
            len(c["c_concat"]) == len(conds[0]["c_concat"]) and all(a is b0 for a, b0 in zip(c["c_concat"], conds[0]["c_concat"]))
            for c in conds[1:])
        shape_sig = (n, tuple(x_noisy.shape), x_noisy.device, share, share and self.cfg_streams)
        st = self._cfg_state
        unet = self.model.diffusion_model
        if st is None or st["key"] is None or st["shape_sig"] != shape_sig or not st["key"].same(cond_t):
            # conditioning changed (new clip): (re)fill the static batch-nB inputs; same-shape buffers are
            # reused so that a captured graph stays valid
            cat = lambda key, cs: torch.cat([torch.cat(c[key], 1) for c in cs], dim=0)
            nx = 1 if share else n                     # copies of the latent-side inputs the UNet is handed
            ctx2, cc2 = cat("c_crossattn", conds), cat("c_concat", conds[:nx]).to(torch.float32)
            fs2 = None if fs is None else torch.cat([fs] * nx, dim=0)
            if st is not None and st["ctx2"].shape == ctx2.shape and st["x2"].shape[1:] == x_noisy.shape[1:] \
                    and st["x2"].shape[0] == nx * b and st["x2"].device == x_noisy.device \
                    and (st["fs2"] is None) == (fs2 is None):
                st["ctx2"].copy_(ctx2)
                st["cc2"].copy_(cc2)
                if fs2 is not None:
                    st["fs2"].copy_(fs2)
            else:
                st = self._cfg_state = dict(
                    ctx2=ctx2.contiguous(), cc2=cc2.contiguous(), fs2=fs2,
                    x2=torch.empty((nx * b, *x_noisy.shape[1:]), dtype=torch.float32, device=x_noisy.device),
                    ts2=torch.empty((nx * b,), dtype=torch.long, device=x_noisy.device), graph=None, calls=0)
            st["key"], st["shape_sig"] = SourceKey(cond_t), shape_sig
            unet.context_cache(st["ctx2"], x_noisy.shape[2])          # project K/V now (in place if cached)
        for k in range(1 if share else n):
            st["x2"][k * b:(k + 1) * b].copy_(x_noisy)
            st["ts2"][k * b:(k + 1) * b].copy_(t)

        branches = share and self.cfg_streams

        def fwd():
            return unet(None, st["ts2"], context=st["ctx2"], fs=st["fs2"], x_parts=[st["x2"], st["cc2"]],
                        replicas=n if share else 1, branches=branches)

        if self.use_hipgraph and x_noisy.is_cuda:
            # a captured graph replays the kernels and the packed-weight pointers it recorded: new weights
            # (load_state_dict / invalidate) or another MXFP8 routing since capture make it stale
            sig = (PackedModule.graph_epoch(), getattr(ops.backend(), "fp8", None), branches)
            if st["graph"] is not None and st.get("graph_sig") != sig:
                st["graph"], st["calls"] = None, 0
            if st["graph"] is None and st["calls"] >= 1:             # first call ran eagerly (warm caches)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    st["out"] = fwd()
                st["graph"], st["graph_sig"] = g, sig
            if st["graph"] is not None:
                st["graph"].replay()
                out = st["out"]
            else:
                out = fwd()
        else:
            out = fwd()
        st["calls"] += 1
        return list(out) if isinstance(out, (list, tuple)) else [out[k * b:(k + 1) * b] for k in range(n)]

    def reset_conditioning(self):
        """Clip boundary (called by the samplers at the start of `sample()`, like one iteration of the
        prompt loop of scripts/evaluation/inference.py:324-342): whatever tensors carry the next
        conditioning, the static batch-nB inputs and every cached K/V projection are re-filled IN PLACE on
        the next guided step (same buffers, so a captured hipGraph stays valid)."""
        if self._cfg_state is not None:
            self._cfg_state["key"] = None
        unet = self.model.diffusion_model
        if hasattr(unet, "reset_conditioning"):
            unet.reset_conditioning()
        dec = getattr(self.first_stage_model, "decoder", None)
        if hasattr(dec, "reset_conditioning"):
            dec.reset_conditioning()

    @torch.no_grad()
    def decode_first_stage(self, z, **kwargs):
        return self.decode_core(z, **kwargs)

    def decode_core(self, z, **kwargs):
        """z: (B, C, T, h, w) latent.  The whole clip batch goes through ONE decoder call with
        timesteps=T (the only geometry in which the dual-reference fusion is well defined for
        B > 1); for B == 1 this is exactly what the reference's
        `perframe_ae=True` loop computes."""
        if z.dim() != 5:
            raise NotImplementedError("decode_first_stage expects a (B, C, T, h, w) video latent")
        ref_context = kwargs.get("ref_context")
        dec = self.first_stage_model.decoder
        scale = 1.0 / float(self.scale_factor)
        # B * T frames share every launch (BASELINE configs[3], the call the reference dies on at ddpm3d.py:656-657).
        # The GEMM kernels address activations block-relatively (csrc/gemm_common.h: tc_tile_row_lo), so a tensor may
        # exceed 2 GiB (level 0 of two 320x512 clips: 2.7 GB); what stays 32-bit is the ROW count of a launch.
        b, _, t, h, w = z.shape
        bmax = max(1, int(0x7fffffff // max(t * (8 * h) * (8 * w), 1)))
        be = ops.backend()
        if getattr(be, "fp8", None) is not None and getattr(be, "fp8_decoder", False):
            # the MXFP8 GEMM still addresses A from the tensor base (csrc/gemm_mx.hip): with TC_FP8_DECODER=1 no
            # activation may exceed the 31-bit byte range -- the widest one is 128 channels at full resolution
            bmax = max(1, min(bmax, int(0x7fffff00 // max(t * (8 * h) * (8 * w) * 128 * 2, 1))))
        if b <= bmax:
            return dec.decode_clip(z, ref_context, scale=scale)
        outs = []
        for i in range(0, b, bmax):
            refs = None if not ref_context else [r[i:i + bmax].contiguous() for r in ref_context]
            outs.append(dec.decode_clip(z[i:i + bmax].contiguous(), refs, scale=scale))
        return torch.cat(outs, 0)


class LatentVisualDiffusion(LatentDiffusion):
    def __init__(self, img_cond_stage_config, image_proj_stage_config, freeze_embedder=True,
                 image_proj_model_trainable=True, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.image_proj_model_trainable = image_proj_model_trainable
        self.embedder = self._instantiate_cond_stage(img_cond_stage_config)
        self.image_proj_model = self._instantiate_cond_stage(image_proj_stage_config)
