"""The caller of the hot path: conditioning -> encode -> DDIM -> two decodes -> splice.

Mirror of reference scripts/evaluation/inference.py: `get_latent_z_with_hidden_states` (164-178) and
`image_guided_synthesis` (180-277) -- same signatures, same returns -- so that a script which imports them
from here instead of defining them runs the whole clip on the MI355X path (embedders, Resampler, first-stage
encoder, sampler, decoder are all the HIP-backed mirrors when the model was built through `dropin`).

One restructuring, result-preserving for everything the function consumes: the reference encodes all T
frames and then keeps only frames 0 and T-1 of both the latent (`img_cat_cond`, :201-204, or frame 0 alone,
:206-207) and the hidden states (:171-175); here only those two frames go through the encoder (8x less
encoder work at T=16).  `z` itself is returned with zeros in the untouched frames; no caller reads them.
"""
from __future__ import annotations

import torch

from .lvdm.ddim import DDIMSampler
from .lvdm.ddim_multiplecond import DDIMSampler as DDIMSampler_multicond


def get_latent_z_with_hidden_states(model, videos):
    """videos (b, c, t, h, w) -> z (b, 4, t, h/8, w/8) with frames 0 and t-1 filled, and the five hidden states
    of the first / last frame as (b, C, 2, H, W) tensors (the decoder's `ref_context`)."""
    b, c, t, h, w = videos.shape
    x = videos[:, :, [0, t - 1]].permute(0, 2, 1, 3, 4).reshape(b * 2, c, h, w)
    posterior, hidden_states = model.first_stage_model.encode(x, return_hidden_states=True)
    hs = [hid.reshape(b, 2, *hid.shape[1:]).permute(0, 2, 1, 3, 4).contiguous() for hid in hidden_states]
    z2 = model.get_first_stage_encoding(posterior).detach()
    z2 = z2.reshape(b, 2, *z2.shape[1:]).permute(0, 2, 1, 3, 4)
    z = torch.zeros((b, z2.shape[1], t, *z2.shape[3:]), dtype=z2.dtype, device=z2.device)
    z[:, :, 0], z[:, :, -1] = z2[:, :, 0], z2[:, :, 1]
    return z, hs


def image_guided_synthesis(model, prompts, videos, noise_shape, n_samples=1, ddim_steps=50, ddim_eta=1.,
                           unconditional_guidance_scale=1.0, cfg_img=None, fs=None, text_input=False,
                           multiple_cond_cfg=False, loop=False, interp=False, timestep_spacing='uniform',
                           guidance_rescale=0.0, **kwargs):
    ddim_sampler = DDIMSampler(model) if not multiple_cond_cfg else DDIMSampler_multicond(model)
    batch_size = noise_shape[0]
    fs = torch.tensor([fs] * batch_size, dtype=torch.long, device=model.device)
    if not text_input:
        prompts = [""] * batch_size

    img = videos[:, :, 0]
    img_emb = model.image_proj_model(model.embedder(img))
    cond_emb = model.get_learned_conditioning(prompts)
    cond = {"c_crossattn": [torch.cat([cond_emb, img_emb], dim=1)]}
    hs = None
    if model.model.conditioning_key == 'hybrid':
        z, hs = get_latent_z_with_hidden_states(model, videos)
        if loop or interp:
            img_cat_cond = torch.zeros_like(z)
            img_cat_cond[:, :, 0] = z[:, :, 0]
            img_cat_cond[:, :, -1] = z[:, :, -1]
        else:
            img_cat_cond = z[:, :, :1].repeat(1, 1, z.shape[2], 1, 1)
        cond["c_concat"] = [img_cat_cond]

    if unconditional_guidance_scale != 1.0:
        if model.uncond_type == "empty_seq":
            uc_emb = model.get_learned_conditioning(batch_size * [""])
        elif model.uncond_type == "zero_embed":
            uc_emb = torch.zeros_like(cond_emb)
        else:
            raise NotImplementedError(model.uncond_type)
        uc_img_emb = model.image_proj_model(model.embedder(torch.zeros_like(img)))
        uc = {"c_crossattn": [torch.cat([uc_emb, uc_img_emb], dim=1)]}
        if model.model.conditioning_key == 'hybrid':
            uc["c_concat"] = [img_cat_cond]
    else:
        uc = None
    additional_decode_kwargs = {'ref_context': hs}

    if multiple_cond_cfg and cfg_img != 1.0:            # one more unconditioning: image = yes, text = ""
        uc_2 = {"c_crossattn": [torch.cat([uc_emb, img_emb], dim=1)]}
        if model.model.conditioning_key == 'hybrid':
            uc_2["c_concat"] = [img_cat_cond]
        kwargs.update({"unconditional_conditioning_img_nonetext": uc_2})
    else:
        kwargs.update({"unconditional_conditioning_img_nonetext": None})

    batch_variants = []
    for _ in range(n_samples):
        samples, _ = ddim_sampler.sample(S=ddim_steps, conditioning=cond, batch_size=batch_size,
                                         shape=noise_shape[1:], verbose=False,
                                         unconditional_guidance_scale=unconditional_guidance_scale,
                                         unconditional_conditioning=uc, eta=ddim_eta, cfg_img=cfg_img, mask=None,
                                         x0=None, fs=fs, timestep_spacing=timestep_spacing,
                                         guidance_rescale=guidance_rescale, **kwargs)
        batch_images = model.decode_first_stage(samples, **additional_decode_kwargs)
        # second decode without frames 1 and T-2; its two centre frames replace the centre of the first (:262-268)
        index = list(range(samples.shape[2]))
        del index[1]
        del index[-2]
        batch_images_middle = model.decode_first_stage(samples[:, :, index], **additional_decode_kwargs)
        mid = batch_images.shape[2] // 2
        batch_images[:, :, mid - 1:mid + 1] = batch_images_middle[:, :, mid - 2:mid]
        batch_variants.append(batch_images)
    batch_variants = torch.stack(batch_variants)            # variants, batch, c, t, h, w
    return batch_variants.permute(1, 0, 2, 3, 4, 5)
