"""One clip, end to end, as three separable stages over the HIP-backed mirror:

    cond    = Conditions.build(model, videos, ...)       # embedders + first / last frame through the encoder
    latents = sample(model, cond, plan)                  # DDIM (two- or three-way guidance)
    video   = decode_spliced(model, latents, cond.refs)  # 16-frame decode + 14-frame re-decode, centre frames spliced

`bench.py` times the second and third stage on resident conditioning; `synthesize` chains all three.  The caller the
reference ships for this path is `image_guided_synthesis` (scripts/evaluation/inference.py:180-277, with its helper
`get_latent_z_with_hidden_states`, :164-178); `image_guided_synthesis` below keeps that call signature (argument names and
order are the interface of the reference's scripts) and is a thin adapter onto the stages.  The unmodified script's own copy
also runs on the mirror (tests/test_dropin_run_cpu.py), so nothing here is needed for the drop-in: this module is the API
for callers that hold conditioning resident and run many clips (bench.py, dist.py).

Result-preserving restructuring: the reference encodes all T frames and keeps frames 0 and T-1 of the latent and of the
hidden states; only those two frames are encoded here (8x less encoder work at T = 16).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import torch

from .lvdm.ddim import DDIMSampler
from .lvdm.ddim_multiplecond import DDIMSampler as ThreeWaySampler


def encode_endpoints(model, videos):
    """(b, c, t, h, w) pixels -> (z_first, z_last) latents of frames 0 and t-1, each (b, 4, h/8, w/8), and the encoder's
    hidden states of those two frames as (b, C, 2, H, W) tensors: the decoder's reference features."""
    b, c, t, h, w = videos.shape
    ends = torch.stack([videos[:, :, 0], videos[:, :, t - 1]], 1).reshape(2 * b, c, h, w)
    posterior, hidden = model.first_stage_model.encode(ends, return_hidden_states=True)
    z = model.get_first_stage_encoding(posterior).detach().unflatten(0, (b, 2))
    refs = [hdn.unflatten(0, (b, 2)).transpose(1, 2).contiguous() for hdn in hidden]
    return z[:, 0], z[:, 1], refs


@dataclass
class Conditions:
    """Everything the sampler and the decoder consume, resident on the device."""
    positive: dict                       # text + image tokens, concat latents
    negative: Optional[dict]             # the unconditional branch (None when guidance is off)
    image_only: Optional[dict]           # third branch of the three-way guidance (image yes, text "")
    refs: Optional[List[torch.Tensor]]   # decoder reference features
    fs: torch.Tensor
    endpoints: Optional[torch.Tensor] = field(default=None, repr=False)    # (b, 4, t, h, w), frames 0 / t-1 filled
    three_way: bool = False              # sampler family (the reference picks it by flag, not by the branch's presence)

    @staticmethod
    def build(model, videos, *, prompts=None, fs=None, guided=True, three_way=False, image_branch=True, hold_endpoints=True):
        """`hold_endpoints`: frames 0 AND t-1 condition the clip (interpolation / looping); otherwise frame 0 is repeated
        over time (plain image-to-video)."""
        b, t = videos.shape[0], videos.shape[2]
        prompts = list(prompts) if prompts is not None else [""] * b
        first = videos[:, :, 0]

        def tokens(text_emb, image):
            return torch.cat([text_emb, model.image_proj_model(model.embedder(image))], dim=1)

        text = model.get_learned_conditioning(prompts)
        pos = {"c_crossattn": [tokens(text, first)]}
        neg = img_only = refs = canvas = None
        hybrid = model.model.conditioning_key == "hybrid"
        if hybrid:
            z0, z1, refs = encode_endpoints(model, videos)
            if hold_endpoints:
                canvas = z0.new_zeros((b, z0.shape[1], t, *z0.shape[2:]))
                canvas[:, :, 0], canvas[:, :, t - 1] = z0, z1
            else:
                canvas = z0.unsqueeze(2).expand(-1, -1, t, -1, -1).contiguous()
            pos["c_concat"] = [canvas]
        if guided:
            if model.uncond_type == "empty_seq":
                blank = model.get_learned_conditioning([""] * b)
            elif model.uncond_type == "zero_embed":
                blank = torch.zeros_like(text)
            else:
                raise NotImplementedError(f"uncond_type {model.uncond_type!r}")
            neg = {"c_crossattn": [tokens(blank, torch.zeros_like(first))]}
            if three_way and image_branch:
                img_only = {"c_crossattn": [torch.cat([blank, pos["c_crossattn"][0][:, text.shape[1]:]], dim=1)]}
            if hybrid:                              # the SAME tensor object: apply_model_multi shares the prefix on it
                for branch in (neg, img_only):
                    if branch is not None:
                        branch["c_concat"] = [canvas]
        fs_t = torch.full((b,), int(fs), dtype=torch.long, device=model.device) if not torch.is_tensor(fs) else fs
        return Conditions(pos, neg, img_only, refs, fs_t, canvas, three_way)


@dataclass
class SamplingPlan:
    steps: int = 50
    eta: float = 1.0
    scale: float = 7.5                     # 1.0 = no guidance
    image_scale: Optional[float] = None    # three-way guidance only
    spacing: str = "uniform_trailing"
    rescale: float = 0.7
    extra: dict = field(default_factory=dict)


def sample(model, cond: Conditions, plan: SamplingPlan, latent_shape, x_T=None):
    """Latents (b, 4, t, h, w) of one clip batch."""
    sampler = (ThreeWaySampler if cond.three_way else DDIMSampler)(model)
    extra = dict(plan.extra)
    x_T = extra.pop("x_T", x_T)                    # the reference's callers hand the start noise over among their **kwargs
    out, _ = sampler.sample(S=plan.steps, conditioning=cond.positive, batch_size=latent_shape[0], shape=tuple(latent_shape[1:]),
                            verbose=False, unconditional_guidance_scale=plan.scale, unconditional_conditioning=cond.negative,
                            eta=plan.eta, cfg_img=plan.image_scale, mask=None, x0=None, fs=cond.fs,
                            timestep_spacing=plan.spacing, guidance_rescale=plan.rescale, x_T=x_T,
                            unconditional_conditioning_img_nonetext=cond.image_only, **extra)
    return out


def decode_spliced(model, latents, refs, marks=None):
    """Decode all T frames; decode again without frames 1 and T-2 (the endpoints' neighbours) and let that second pass's two
    centre frames replace the first pass's (the reference's anti-flicker step, inference.py:262-268).  `marks`: optional
    callable invoked between the passes (bench.py records HIP events there)."""
    t = latents.shape[2]
    video = model.decode_first_stage(latents, ref_context=refs)
    if marks is not None:
        marks()
    keep = [i for i in range(t) if i not in (1, t - 2)]
    again = model.decode_first_stage(latents[:, :, keep].contiguous(), ref_context=refs)
    c = t // 2
    video[:, :, c - 1:c + 1] = again[:, :, c - 2:c]
    return video


def synthesize(model, videos, latent_shape, *, plan: SamplingPlan, prompts=None, fs=24, three_way=False, hold_endpoints=True,
               variants=1):
    """(b, variants, 3, t, H, W) pixels in [-1, 1]."""
    cond = Conditions.build(model, videos, prompts=prompts, fs=fs, guided=plan.scale != 1.0,
                            three_way=three_way, image_branch=plan.image_scale != 1.0, hold_endpoints=hold_endpoints)
    clips = [decode_spliced(model, sample(model, cond, plan, latent_shape), cond.refs) for _ in range(variants)]
    return torch.stack(clips, dim=1)


# --- the reference scripts' call signatures (scripts/evaluation/inference.py:164,180) -----------------------------------

def get_latent_z_with_hidden_states(model, videos):
    z0, z1, refs = encode_endpoints(model, videos)
    b, t = videos.shape[0], videos.shape[2]
    z = z0.new_zeros((b, z0.shape[1], t, *z0.shape[2:]))
    z[:, :, 0], z[:, :, t - 1] = z0, z1
    return z, refs


def image_guided_synthesis(model, prompts, videos, noise_shape, n_samples=1, ddim_steps=50, ddim_eta=1.,
                           unconditional_guidance_scale=1.0, cfg_img=None, fs=None, text_input=False,
                           multiple_cond_cfg=False, loop=False, interp=False, timestep_spacing='uniform',
                           guidance_rescale=0.0, **kwargs):
    plan = SamplingPlan(steps=ddim_steps, eta=ddim_eta, scale=unconditional_guidance_scale, image_scale=cfg_img,
                        spacing=timestep_spacing, rescale=guidance_rescale, extra=kwargs)
    return synthesize(model, videos, noise_shape, plan=plan, prompts=prompts if text_input else None, fs=fs,
                      three_way=multiple_cond_cfg, hold_endpoints=bool(loop or interp), variants=n_samples)
