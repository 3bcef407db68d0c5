"""DDIM sampler with separate text / image guidance.

Interface of reference lvdm/models/samplers/ddim_multiplecond.py: the same class as
samplers/ddim.py (make_schedule, sample, ddim_sampling are line-for-line the same there) with a
`p_sample_ddim` (210-288) that runs THREE UNet passes per step -- full condition, unconditional,
and "image kept, text dropped" (`kwargs['unconditional_conditioning_img_nonetext']`) -- and combines

    e = e_uncond + cfg_img * (e_uncond_img - e_uncond) + s * (e_cond - e_uncond_img)        (:236)

before the usual guidance rescale against e_cond, v-parameterisation and DDIM update.
scripts/evaluation/funcs.py:61-76 selects it with `multiple_cond_cfg=True`.

Here the three passes are one batch-3B UNet call (`apply_model_multi`) and the combination is part
of the fused tc_ddim_step kernel (TcDdimParams.e_uncond_img / cfg_img).
"""
from __future__ import annotations

import torch

from .. import ops
from . import ddim as _ddim
from .ddim import DDIMSampler as _DDIMSampler


class DDIMSampler(_DDIMSampler):
    @torch.no_grad()
    def p_sample_ddim(self, x, c, t, index, repeat_noise=False, use_original_steps=False,
                      quantize_denoised=False, temperature=1., noise_dropout=0., score_corrector=None,
                      corrector_kwargs=None, unconditional_guidance_scale=1., unconditional_conditioning=None,
                      uc_type=None, cfg_img=None, mask=None, x0=None, guidance_rescale=0.0, _step=None, **kwargs):
        if use_original_steps or quantize_denoised or score_corrector is not None or noise_dropout > 0.:
            raise NotImplementedError("p_sample_ddim variant unused by the inference scripts")
        if self.model.parameterization != "v":
            raise NotImplementedError("the fused DDIM step implements the v-parameterisation of the config")
        if cfg_img is None:
            cfg_img = unconditional_guidance_scale                                  # :217-218
        uc_img = kwargs['unconditional_conditioning_img_nonetext']                   # :220 (KeyError like the reference)
        step = int(t[0]) if _step is None else _step
        use_cfg = not (unconditional_conditioning is None or unconditional_guidance_scale == 1.)
        e_u = e_i = None
        if not use_cfg:
            e_c = self.model.apply_model(x, t, c, **kwargs)
        elif hasattr(self.model, "apply_model_multi"):
            e_c, e_u, e_i = self.model.apply_model_multi(x, t, [c, unconditional_conditioning, uc_img], **kwargs)
        else:
            e_c = self.model.apply_model(x, t, c, **kwargs)
            e_u = self.model.apply_model(x, t, unconditional_conditioning, **kwargs)
            e_i = self.model.apply_model(x, t, uc_img, **kwargs)
        sc = self.step_scalars(index, step)
        noise = _ddim.noise_like(x.shape, x.device, repeat_noise)     # always drawn, like the reference (:277)
        if sc["sigma"] != 0.0:
            if temperature != 1.:
                noise = noise * temperature
            noise = noise.to(torch.float32).contiguous()
        else:
            noise = None
        cont = lambda v: None if v is None else v.contiguous()
        return ops.ddim_step(x.contiguous(), e_c.contiguous(), cont(e_u), noise,
                             cfg_scale=unconditional_guidance_scale,
                             guidance_rescale=guidance_rescale if use_cfg else 0.0,
                             e_uncond_img=cont(e_i), cfg_img=cfg_img, **sc)
