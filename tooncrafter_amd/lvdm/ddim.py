"""DDIM sampler driving the HIP UNet.

Interface of reference lvdm/models/samplers/ddim.py: DDIMSampler.__init__ (11-16),
register_buffer (18-22), make_schedule (24-57), sample (60-132), ddim_sampling
(135-203), p_sample_ddim (206-279) -- same signatures, kwargs and return values, so
scripts/evaluation/inference.py:244-259 and funcs.py:61-76 call it unchanged.

What is different underneath:
  * the per-step algebra (CFG combine, guidance rescale with its two unbiased
    stds, v -> eps/x0, dynamic rescale, x_prev) is ONE fused kernel (tc_ddim_step);
  * its six scalars per step are computed on the host from the schedule tables in the
    reference's exact order/precision, so the loop has no device->host sync (the
    reference reads six device scalars per step, ddim.py:251-264);
  * conditional and unconditional UNet passes run as one B=2 call when the model
    offers `apply_model_cfg` (every normalisation in the UNet is per-sample, so this is
    exact), reading the 2.9 GB of weights once per step instead of twice.
"""
from __future__ import annotations

import numpy as np
import torch
from tqdm import tqdm

from .. import ops
from .utils_diffusion import make_ddim_sampling_parameters, make_ddim_timesteps


def noise_like(shape, device, repeat=False):
    """Gaussian draw from the device generator (reference lvdm/common.py:31-34).  Module-level
    name so that parity tests can substitute an injected noise source, as they do for the
    reference (`lvdm.models.samplers.ddim.noise_like`)."""
    if repeat:
        return torch.randn((1, *shape[1:]), device=device).repeat(shape[0], *((1,) * (len(shape) - 1)))
    return torch.randn(shape, device=device)


class DDIMSampler(object):
    def __init__(self, model, schedule="linear", **kwargs):
        super().__init__()
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule
        self.counter = 0

    def register_buffer(self, name, attr):
        if isinstance(attr, torch.Tensor) and attr.device != self.model.device:
            attr = attr.to(self.model.device)        # follow the model (the reference hard-codes "cuda")
        setattr(self, name, attr)

    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=True):
        self.ddim_timesteps = make_ddim_timesteps(ddim_discr_method=ddim_discretize,
                                                  num_ddim_timesteps=ddim_num_steps,
                                                  num_ddpm_timesteps=self.ddpm_num_timesteps, verbose=verbose)
        m = self.model
        ac = m.alphas_cumprod.detach().to(torch.float32).cpu()
        assert ac.shape[0] == self.ddpm_num_timesteps, 'alphas have to be defined for each timestep'
        if m.use_dynamic_rescale:
            sa = m.scale_arr.detach().to(torch.float32).cpu()[torch.as_tensor(self.ddim_timesteps, dtype=torch.long)]
            self.ddim_scale_arr = sa
            self.ddim_scale_arr_prev = torch.cat([sa[0:1], sa[:-1]])
        self.register_buffer('betas', m.betas.detach().to(torch.float32))
        self.register_buffer('alphas_cumprod', m.alphas_cumprod.detach().to(torch.float32))
        self.register_buffer('alphas_cumprod_prev', m.alphas_cumprod_prev.detach().to(torch.float32))
        sigmas, alphas, alphas_prev = make_ddim_sampling_parameters(ac, self.ddim_timesteps, ddim_eta, verbose)
        self.ddim_sigmas, self.ddim_alphas, self.ddim_alphas_prev = sigmas, alphas, alphas_prev
        self.ddim_sqrt_one_minus_alphas = np.sqrt(1. - alphas)
        # host copies of the two per-timestep fp32 buffers the v-parameterisation reads
        self._sqrt_ac = m.sqrt_alphas_cumprod.detach().to(torch.float32).cpu()
        self._sqrt_1m_ac = m.sqrt_one_minus_alphas_cumprod.detach().to(torch.float32).cpu()

    def step_scalars(self, index: int, t: int):
        """The fp32 scalars of one update, each rounded exactly where the reference's
        `torch.full(size, value)` rounds it (ddim.py:251-266, 271, 277)."""
        f32 = lambda v: torch.tensor(float(v), dtype=torch.float32)
        a_prev = f32(self.ddim_alphas_prev[index])
        sigma = f32(self.ddim_sigmas[index])
        dir_coef = (1. - a_prev - sigma ** 2).sqrt()
        x0_rescale = f32(1.0)
        if self.model.use_dynamic_rescale:
            x0_rescale = self.ddim_scale_arr_prev[index] / self.ddim_scale_arr[index]
        return dict(sqrt_ac=float(self._sqrt_ac[t]), sqrt_1m_ac=float(self._sqrt_1m_ac[t]),
                    sqrt_a_prev=float(a_prev.sqrt()), dir_coef=float(dir_coef), sigma=float(sigma),
                    x0_rescale=float(x0_rescale))

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None,
               img_callback=None, quantize_x0=False, eta=0., mask=None, x0=None, temperature=1.,
               noise_dropout=0., score_corrector=None, corrector_kwargs=None, verbose=True,
               schedule_verbose=False, x_T=None, log_every_t=100, unconditional_guidance_scale=1.,
               unconditional_conditioning=None, precision=None, fs=None, timestep_spacing='uniform',
               guidance_rescale=0.0, **kwargs):
        if conditioning is not None:
            if isinstance(conditioning, dict):
                first = conditioning[list(conditioning.keys())[0]]
                cbs = (first[0] if isinstance(first, (list, tuple)) else first).shape[0]
            else:
                cbs = conditioning.shape[0]
            if cbs != batch_size:
                print(f"Warning: Got {cbs} conditionings but batch-size is {batch_size}")
        self.make_schedule(ddim_num_steps=S, ddim_discretize=timestep_spacing, ddim_eta=eta,
                           verbose=schedule_verbose)
        if hasattr(self.model, "reset_conditioning"):
            self.model.reset_conditioning()          # a sampling run = one clip: never reuse cached conditioning
        if len(shape) == 3:
            size = (batch_size, *shape)
        else:
            c, t, h, w = shape
            size = (batch_size, c, t, h, w)
        return self.ddim_sampling(conditioning, size, callback=callback, img_callback=img_callback,
                                  quantize_denoised=quantize_x0, mask=mask, x0=x0,
                                  ddim_use_original_steps=False, noise_dropout=noise_dropout,
                                  temperature=temperature, score_corrector=score_corrector,
                                  corrector_kwargs=corrector_kwargs, x_T=x_T, log_every_t=log_every_t,
                                  unconditional_guidance_scale=unconditional_guidance_scale,
                                  unconditional_conditioning=unconditional_conditioning, verbose=verbose,
                                  precision=precision, fs=fs, guidance_rescale=guidance_rescale, **kwargs)

    @torch.no_grad()
    def ddim_sampling(self, cond, shape, x_T=None, ddim_use_original_steps=False, callback=None,
                      timesteps=None, quantize_denoised=False, mask=None, x0=None, img_callback=None,
                      log_every_t=100, temperature=1., noise_dropout=0., score_corrector=None,
                      corrector_kwargs=None, unconditional_guidance_scale=1., unconditional_conditioning=None,
                      verbose=True, precision=None, fs=None, guidance_rescale=0.0, **kwargs):
        if ddim_use_original_steps or timesteps is not None or quantize_denoised or score_corrector is not None:
            raise NotImplementedError("only the DDIM-subsequence sampling path the scripts use")
        if mask is not None:
            raise NotImplementedError("mask blending (ddim.py:174-180) is unused by the interpolation scripts")
        device = self.model.betas.device
        b = shape[0]
        img = torch.randn(shape, device=device) if x_T is None else x_T
        img = img.to(torch.float32).contiguous()
        steps = self.ddim_timesteps
        total_steps = steps.shape[0]
        intermediates = {'x_inter': [img], 'pred_x0': [img]}
        time_range = np.flip(steps)
        iterator = tqdm(time_range, desc='DDIM Sampler', total=total_steps) if verbose else time_range
        kwargs.pop("clean_cond", False)
        for i, step in enumerate(iterator):
            index = total_steps - i - 1
            ts = torch.full((b,), int(step), device=device, dtype=torch.long)
            img, pred_x0 = self.p_sample_ddim(img, cond, ts, index=index, temperature=temperature,
                                              noise_dropout=noise_dropout,
                                              unconditional_guidance_scale=unconditional_guidance_scale,
                                              unconditional_conditioning=unconditional_conditioning,
                                              fs=fs, guidance_rescale=guidance_rescale, _step=int(step), **kwargs)
            if callback:
                callback(i)
            if img_callback:
                img_callback(pred_x0, i)
            if index % log_every_t == 0 or index == total_steps - 1:
                intermediates['x_inter'].append(img)
                intermediates['pred_x0'].append(pred_x0)
        return img, intermediates

    @torch.no_grad()
    def p_sample_ddim(self, x, c, t, index, repeat_noise=False, use_original_steps=False,
                      quantize_denoised=False, temperature=1., noise_dropout=0., score_corrector=None,
                      corrector_kwargs=None, unconditional_guidance_scale=1., unconditional_conditioning=None,
                      uc_type=None, conditional_guidance_scale_temporal=None, mask=None, x0=None,
                      guidance_rescale=0.0, _step=None, **kwargs):
        if use_original_steps or quantize_denoised or score_corrector is not None or noise_dropout > 0.:
            raise NotImplementedError("p_sample_ddim variant unused by the inference scripts")
        if self.model.parameterization != "v":
            raise NotImplementedError("the fused DDIM step implements the v-parameterisation of the config")
        step = int(t[0]) if _step is None else _step
        use_cfg = not (unconditional_conditioning is None or unconditional_guidance_scale == 1.)
        if not use_cfg:
            e_c, e_u = self.model.apply_model(x, t, c, **kwargs), None
        elif hasattr(self.model, "apply_model_cfg"):
            e_c, e_u = self.model.apply_model_cfg(x, t, c, unconditional_conditioning, **kwargs)
        else:
            e_c = self.model.apply_model(x, t, c, **kwargs)
            e_u = self.model.apply_model(x, t, unconditional_conditioning, **kwargs)
        sc = self.step_scalars(index, step)
        # the reference draws the noise every step, also when sigma == 0 (ddim.py:266): draw (and drop) it so
        # the device generator stays in step with the reference for later draws (x_T of the next variant)
        noise = noise_like(x.shape, x.device, repeat_noise)
        if sc["sigma"] != 0.0:
            if temperature != 1.:
                noise = noise * temperature
            noise = noise.to(torch.float32).contiguous()
        else:
            noise = None
        x_prev, pred_x0 = ops.ddim_step(x.contiguous(), e_c.contiguous(), None if e_u is None else e_u.contiguous(),
                                        noise, cfg_scale=unconditional_guidance_scale,
                                        guidance_rescale=guidance_rescale if use_cfg else 0.0, **sc)
        return x_prev, pred_x0
