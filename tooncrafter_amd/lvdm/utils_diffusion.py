"""Noise schedule helpers of the sampler (host side, numpy/torch CPU scalars).

Interface of reference lvdm/models/utils_diffusion.py: make_beta_schedule (31-53),
make_ddim_timesteps (56-76), make_ddim_sampling_parameters (79-91),
rescale_zero_terminal_snr (112-144).  These run once per sampling call on 1000- or
50-entry tables; the per-element math (timestep_embedding, rescale_noise_cfg) lives
in the HIP kernels tc_timestep_embedding / tc_ddim_step.
"""
from __future__ import annotations

import numpy as np
import torch


def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
    if schedule != "linear":
        raise NotImplementedError(f"beta schedule '{schedule}' (the ToonCrafter config uses 'linear')")
    lo, hi = linear_start ** 0.5, linear_end ** 0.5
    return (torch.linspace(lo, hi, n_timestep, dtype=torch.float64, device="cpu") ** 2).numpy()


def rescale_zero_terminal_snr(betas):
    """Shift/scale sqrt(alpha_bar) so that the last timestep has exactly zero SNR
    (arXiv 2305.08891, Algorithm 1)."""
    s = np.sqrt(np.cumprod(1.0 - betas, axis=0))
    s0, sT = s[0].copy(), s[-1].copy()
    s = (s - sT) * (s0 / (s0 - sT))
    abar = s ** 2
    alphas = np.concatenate([abar[0:1], abar[1:] / abar[:-1]])
    return 1 - alphas


def make_ddim_timesteps(ddim_discr_method, num_ddim_timesteps, num_ddpm_timesteps, verbose=True):
    if ddim_discr_method == "uniform":
        c = num_ddpm_timesteps // num_ddim_timesteps
        steps = np.arange(0, num_ddpm_timesteps, c) + 1
    elif ddim_discr_method == "uniform_trailing":
        c = num_ddpm_timesteps / num_ddim_timesteps
        steps = np.flip(np.round(np.arange(num_ddpm_timesteps, 0, -c))).astype(np.int64) - 1
    elif ddim_discr_method == "quad":
        steps = ((np.linspace(0, np.sqrt(num_ddpm_timesteps * .8), num_ddim_timesteps)) ** 2).astype(int) + 1
    else:
        raise NotImplementedError(f'There is no ddim discretization method called "{ddim_discr_method}"')
    if verbose:
        print(f"Selected timesteps for ddim sampler: {steps}")
    return steps


def make_ddim_sampling_parameters(alphacums, ddim_timesteps, eta, verbose=True):
    """sigma_t = eta * sqrt((1-a_prev)/(1-a_t) * (1 - a_t/a_prev)) (arXiv 2010.02502).

    Evaluated with the reference's precision pattern, which matters at the singular first
    step of a zero-terminal-SNR schedule: alphas are fp32, `1/(1 - a_t)` is evaluated in fp32,
    everything else is float64 on those fp32 values.  Returns (sigmas f64, alphas f32,
    alphas_prev f64) as numpy arrays."""
    ac = torch.as_tensor(alphacums).detach().to(torch.float32).cpu()
    alphas = ac[torch.as_tensor(np.asarray(ddim_timesteps), dtype=torch.long)]
    alphas_prev = np.asarray([float(ac[0])] + [float(v) for v in alphas[:-1]], dtype=np.float64)
    # the reference's `ndarray / tensor` dispatches to Tensor.__rtruediv__ = reciprocal() * other,
    # i.e. 1/(1 - a_t) is an fp32 reciprocal of an fp32 subtraction
    inv_one_minus_a = (1 - alphas).reciprocal().to(torch.float64).numpy()
    ratio = alphas.to(torch.float64).numpy() / alphas_prev
    sigmas = eta * np.sqrt(inv_one_minus_a * (1 - alphas_prev) * (1 - ratio))
    return sigmas, alphas.numpy(), alphas_prev
