"""Deterministic stand-ins for the three conditioners, shared by make_golden.py (plugged into the REFERENCE model)
and tests (plugged into the mirror), so that the `image_guided_synthesis` golden pins the orchestration only:
conditioning assembly, first/last-frame latents and hidden states, the sampler call, the two decodes, the splice."""
import torch
import torch.nn as nn
import torch.nn.functional as F

CTX, N_TEXT, N_IMG_TOK = 96, 77, 16


def _fixed(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


class StubEmbedder(nn.Module):
    """image (B, 3, H, W) -> (B, 5, 48): pooled pixels times fixed matrices."""

    def forward(self, img):
        f = F.adaptive_avg_pool2d(img.float().cpu(), (4, 4)).reshape(img.shape[0], 1, 48)
        return (f * (1.0 + 0.1 * torch.arange(5.0).view(1, 5, 1)) + 0.05 * _fixed((1, 5, 48), 1)).to(img.device)


class StubImageProj(nn.Module):
    """(B, 5, 48) -> (B, 16 * T, CTX)"""

    def __init__(self, t):
        super().__init__()
        self.t = t

    def forward(self, x):
        w = _fixed((48, CTX), 2) * 48 ** -0.5
        v = (x.float().cpu().mean(1) @ w)[:, None, :]                           # (B, 1, CTX)
        return (v * _fixed((1, N_IMG_TOK * self.t, 1), 3) + 0.3 * _fixed((1, N_IMG_TOK * self.t, CTX), 4)).to(x.device)


def stub_text(prompts, device="cpu"):
    """any prompt list -> (B, 77, CTX); the empty prompt and a non-empty one differ"""
    out = [_fixed((1, N_TEXT, CTX), 5 + (sum(map(ord, p)) % 1000)) for p in prompts]
    return torch.cat(out, 0).to(device)
