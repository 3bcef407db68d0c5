"""Synthetic weights and inputs for the ToonCrafter hot path.

There is no network on the build or GPU boxes, so every test and bench runs on
synthetic weights.  The reference's own default init is unusable for parity
work: it zero-initialises the output conv of the UNet and the last layer of
every residual branch (reference lvdm/modules/networks/openaimodel3d.py:179,545,
269-270, 381-382; lvdm/modules/attention.py:288-290,360-362), so a "random
init" checkpoint produces an identically-zero UNet output.

The recipe here is a pure function of (parameter name, shape, seed): every
tensor is drawn from its own generator seeded by crc32(name) ^ seed.  That makes
the same state dict reproducible on any box, for the reference modules (golden
generation), the CPU oracle and the HIP path alike, without serialising 1.4 B
parameters.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Iterable, Tuple

import torch


def _seed_for(name: str, seed: int) -> int:
    return (zlib.crc32(name.encode("utf-8")) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF


def synth_tensor(name: str, shape: Tuple[int, ...], seed: int = 1234,
                 device: str | torch.device = "cpu") -> torch.Tensor:
    """One synthetic parameter, fp32.

    * ``mix_factor`` scalars (VideoResBlock blend, reference
      autoencoder_dualref.py:875-882): N(0, 0.5) so sigmoid(mix) is not 0.5 exactly.
    * 1-D ``weight`` (GroupNorm / LayerNorm gamma): 1 + 0.1 N(0,1).
    * 1-D ``bias``: 0.05 N(0,1).
    * >=2-D weights: N(0, 1/(3 fan_in)), the variance of torch's default
      kaiming-uniform(a=sqrt(5)) init, including the tensors the reference zeroes.
    * schedule buffers are not parameters and never come through here.
    """
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(_seed_for(name, seed))
    shape = tuple(int(s) for s in shape)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "mix_factor":
        return 0.5 * torch.randn(shape, generator=g, device=dev)
    if leaf == "alpha":
        return 0.1 * torch.randn(shape, generator=g, device=dev)
    if len(shape) <= 1:
        r = torch.randn(shape, generator=g, device=dev)
        if leaf == "weight":
            return 1.0 + 0.1 * r
        return 0.05 * r
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    std = 1.0 / math.sqrt(3.0 * fan_in)
    return std * torch.randn(shape, generator=g, device=dev)


def synth_state_dict(shapes: Dict[str, Tuple[int, ...]] | Iterable[Tuple[str, Tuple[int, ...]]],
                     seed: int = 1234, device: str | torch.device = "cpu",
                     dtype: torch.dtype = torch.float32) -> Dict[str, torch.Tensor]:
    items = shapes.items() if isinstance(shapes, dict) else shapes
    return {k: synth_tensor(k, tuple(s), seed, device).to(dtype) for k, s in items}


def fill_module_(module: torch.nn.Module, seed: int = 1234, prefix: str = "") -> None:
    """Overwrite every parameter of ``module`` in place with its synthetic value.

    Generated on CPU (so values match the oracle bit for bit) and copied to the
    parameter's device.  Buffers (schedules) are left alone.
    """
    with torch.no_grad():
        for name, p in module.named_parameters():
            v = synth_tensor(prefix + name, tuple(p.shape), seed, "cpu")
            p.copy_(v.to(p.dtype))


def synth_inputs(batch: int, t: int, h: int, w: int, context_dim: int = 1024,
                 n_img_tokens_per_frame: int = 16, text_len: int = 77,
                 seed: int = 7, z_channels: int = 4) -> Dict[str, torch.Tensor]:
    """Synthetic sampler inputs with the shapes of reference
    scripts/evaluation/inference.py:189-243: x_T, the hybrid ``c_concat`` latent
    (encoded first/last frame, zeros elsewhere), cond/uncond cross-attention
    context ``[B, 77 + 16*T, C]`` and the ``fs`` frame-stride token."""
    g = torch.Generator().manual_seed(seed)
    x_T = torch.randn(batch, z_channels, t, h, w, generator=g)
    z = torch.randn(batch, z_channels, 2, h, w, generator=g) * (0.18215 * 4.0)
    c_concat = torch.zeros(batch, z_channels, t, h, w)
    c_concat[:, :, 0] = z[:, :, 0]
    c_concat[:, :, -1] = z[:, :, 1]
    L = text_len + n_img_tokens_per_frame * t
    cond = torch.randn(batch, L, context_dim, generator=g)
    uncond = torch.randn(batch, L, context_dim, generator=g)
    fs = torch.full((batch,), 10, dtype=torch.long)
    return {"x_T": x_T, "c_concat": c_concat, "cond": cond, "uncond": uncond, "fs": fs}


def synth_ref_context(batch: int, h: int, w: int, ch: int = 128,
                      ch_mult=(1, 2, 4, 4), seed: int = 11):
    """Five encoder hidden states of the first/last frame, ``(B, C, 2, H, W)``
    each, ordered like reference ae_modules.py:441-458 returns them: one per
    decoder level (level 0 = full resolution) plus the pre-``conv_out`` one.
    ``h, w`` are the latent sizes; level i lives at ``h * 2**(L-1-i)``."""
    g = torch.Generator().manual_seed(seed)
    L = len(ch_mult)
    out = []
    for lvl in range(L):
        s = 2 ** (L - 1 - lvl)
        out.append(torch.randn(batch, ch * ch_mult[lvl], 2, h * s, w * s, generator=g))
    s = 2 ** (L - 1)
    out.append(torch.randn(batch, ch * ch_mult[0], 2, h * s, w * s, generator=g))
    return out
