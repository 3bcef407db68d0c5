"""Oracle: noise schedule, DDIM tables, CFG / rescale / v-param algebra, DDIM loop.

Restates (reference paths under lvdm/):
  models/utils_diffusion.py:31-35     make_beta_schedule('linear')
  models/utils_diffusion.py:112-144   rescale_zero_terminal_snr
  models/ddpm3d.py:124-187            DDPM.register_schedule (fp32 buffers)
  models/ddpm3d.py:523-528            dynamic-rescale scale_arr
  models/utils_diffusion.py:56-76     make_ddim_timesteps
  models/utils_diffusion.py:79-91     make_ddim_sampling_parameters
  models/samplers/ddim.py:24-57       DDIMSampler.make_schedule
  models/samplers/ddim.py:135-203     ddim_sampling
  models/samplers/ddim.py:206-279     p_sample_ddim
  models/samplers/ddim_multiplecond.py:210-288   p_sample_ddim with three-way guidance (row f3)
  models/utils_diffusion.py:147-158   rescale_noise_cfg (unbiased std)
  models/ddpm3d.py:240-252            predict_start_from_z_and_v / predict_eps_from_z_and_v

The reference mixes dtypes here (float64 numpy `ddim_alphas_prev`/`ddim_sigmas`,
float32 torch `ddim_alphas`, scalars re-materialised as fp32 by `torch.full`);
this file reproduces that order of operations exactly, because at the first
step (index S-1, zero terminal SNR) the radicand 1 - a_prev - sigma^2 is
+5.96e-8 in the reference's order and can go negative in any other.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import numpy as np
import torch


def make_schedule_buffers(timesteps: int = 1000, linear_start: float = 0.00085,
                          linear_end: float = 0.012, rescale_zero_snr: bool = True,
                          use_dynamic_rescale: bool = True, base_scale: float = 0.7,
                          turning_step: int = 400) -> Dict[str, torch.Tensor]:
    betas = (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=torch.float64) ** 2).numpy()
    if rescale_zero_snr:
        abar_sqrt = np.sqrt(np.cumprod(1.0 - betas, axis=0))
        a0, aT = abar_sqrt[0].copy(), abar_sqrt[-1].copy()
        abar_sqrt -= aT
        abar_sqrt *= a0 / (a0 - aT)
        abar = abar_sqrt ** 2
        alphas = np.concatenate([abar[0:1], abar[1:] / abar[:-1]])
        betas = 1 - alphas
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    f32 = lambda a: torch.tensor(a, dtype=torch.float32)
    out = {
        "betas": f32(betas),
        "alphas_cumprod": f32(ac),
        "alphas_cumprod_prev": f32(ac_prev),
        "sqrt_alphas_cumprod": f32(np.sqrt(ac)),
        "sqrt_one_minus_alphas_cumprod": f32(np.sqrt(1.0 - ac)),
    }
    if use_dynamic_rescale:
        out["scale_arr"] = f32(np.concatenate((np.linspace(1.0, base_scale, turning_step),
                                               np.full(timesteps, base_scale))))
    return out


def make_ddim_timesteps(method: str, S: int, T: int = 1000) -> np.ndarray:
    if method == "uniform":
        c = T // S
        return np.asarray(list(range(0, T, c))) + 1
    if method == "uniform_trailing":
        c = T / S
        return np.flip(np.round(np.arange(T, 0, -c))).astype(np.int64) - 1
    if method == "quad":
        return ((np.linspace(0, np.sqrt(T * .8), S)) ** 2).astype(int) + 1
    raise NotImplementedError(method)


def make_ddim_tables(buffers: Dict[str, torch.Tensor], S: int, eta: float,
                     method: str = "uniform_trailing") -> Dict[str, object]:
    """The per-index tables `p_sample_ddim` reads, with the reference's dtypes."""
    ts = make_ddim_timesteps(method, S, buffers["alphas_cumprod"].shape[0])
    ac = buffers["alphas_cumprod"].to(torch.float32)          # to_torch(...) at ddim.py:36
    alphas = ac[ts]                                           # float32 torch
    alphas_prev = np.asarray([ac[0]] + ac[ts[:-1]].tolist())  # float64 numpy of fp32 values
    sigmas = eta * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))  # float64 torch
    tab = {
        "timesteps": ts,
        "alphas": alphas,
        "alphas_prev": alphas_prev,
        "sigmas": sigmas,
        "sqrt_one_minus_alphas": np.sqrt(1.0 - alphas),
    }
    if "scale_arr" in buffers:
        sa = buffers["scale_arr"][ts]
        tab["scale_arr"] = sa
        tab["scale_arr_prev"] = torch.cat([sa[0:1], sa[:-1]])
    return tab


def rescale_noise_cfg(cfg: torch.Tensor, pred_text: torch.Tensor, guidance_rescale: float) -> torch.Tensor:
    dims = list(range(1, cfg.ndim))
    std_text = pred_text.std(dim=dims, keepdim=True)
    std_cfg = cfg.std(dim=dims, keepdim=True)
    return guidance_rescale * (cfg * (std_text / std_cfg)) + (1 - guidance_rescale) * cfg


def ddim_step(x: torch.Tensor, e_cond: torch.Tensor, e_uncond: Optional[torch.Tensor], t: int, index: int,
              tab: Dict[str, object], buffers: Dict[str, torch.Tensor], cfg_scale: float,
              guidance_rescale: float, noise: Optional[torch.Tensor],
              e_uncond_img: Optional[torch.Tensor] = None, cfg_img: Optional[float] = None):
    """One p_sample_ddim update (v-parameterisation, dynamic rescale).
    With `e_uncond_img` (UNet pass that keeps the image condition and drops the text) the guidance
    is the three-way form of ddim_multiplecond.py:226-236; `cfg_img=None` means `cfg_scale` (:217-218).
    Returns (x_prev, pred_x0)."""
    b = x.shape[0]
    size = (b,) + (1,) * (x.dim() - 1)
    if e_uncond is None or cfg_scale == 1.0:
        out = e_cond
    else:
        if e_uncond_img is not None:
            ci = cfg_scale if cfg_img is None else cfg_img
            out = e_uncond + ci * (e_uncond_img - e_uncond) + cfg_scale * (e_cond - e_uncond_img)
        else:
            out = e_uncond + cfg_scale * (e_cond - e_uncond)
        if guidance_rescale > 0.0:
            out = rescale_noise_cfg(out, e_cond, guidance_rescale)
    sa = buffers["sqrt_alphas_cumprod"][t]
    s1 = buffers["sqrt_one_minus_alphas_cumprod"][t]
    e_t = sa * out + s1 * x
    pred_x0 = sa * x - s1 * out
    full = lambda v: torch.full(size, float(v), dtype=torch.float32, device=x.device)
    a_prev = full(tab["alphas_prev"][index])
    sigma_t = full(tab["sigmas"][index])
    if "scale_arr" in tab:
        pred_x0 = pred_x0 * (full(tab["scale_arr_prev"][index]) / full(tab["scale_arr"][index]))
    dir_xt = (1.0 - a_prev - sigma_t ** 2).sqrt() * e_t
    x_prev = a_prev.sqrt() * pred_x0 + dir_xt
    if noise is not None:
        x_prev = x_prev + sigma_t * noise
    return x_prev, pred_x0


def ddim_sample(apply_model: Callable, x_T: torch.Tensor, cond, uncond, S: int, eta: float,
                cfg_scale: float, guidance_rescale: float, buffers: Dict[str, torch.Tensor],
                method: str = "uniform_trailing",
                noise_fn: Optional[Callable[[int], torch.Tensor]] = None,
                step_callback: Optional[Callable] = None,
                uncond_img=None, cfg_img: Optional[float] = None):
    """ddim.py:135-203.  `apply_model(x, t_long[b], c)` is the UNet call;
    `noise_fn(i)` supplies the step-i Gaussian draw (the reference takes it from
    the device generator, which cannot be matched across devices, so parity runs
    inject it)."""
    tab = make_ddim_tables(buffers, S, eta, method)
    img = x_T
    b = x_T.shape[0]
    for i, step in enumerate(np.flip(tab["timesteps"])):
        index = S - i - 1
        ts = torch.full((b,), int(step), dtype=torch.long, device=x_T.device)
        e_c = apply_model(img, ts, cond)
        e_u = apply_model(img, ts, uncond) if (uncond is not None and cfg_scale != 1.0) else None
        e_i = apply_model(img, ts, uncond_img) if (e_u is not None and uncond_img is not None) else None
        noise = noise_fn(i) if (noise_fn is not None and eta > 0) else None
        img, pred_x0 = ddim_step(img, e_c, e_u, int(step), index, tab, buffers, cfg_scale,
                                 guidance_rescale, noise, e_uncond_img=e_i, cfg_img=cfg_img)
        if step_callback is not None:
            step_callback(i, img, pred_x0)
    return img
