"""Shared host-side plumbing of the module mirror: the channels-last activation
handle and the weight packers that turn reference-layout parameters into the
bf16 `[N, K]` matrices the HIP GEMM consumes."""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.nn as nn

from .. import ops

BF16 = torch.bfloat16


def ceil_to(x: int, m: int) -> int:
    return (x + m - 1) // m * m


@dataclass
class Act:
    """A (B, T, H, W, C) activation stored as bf16 rows [B*T*H*W, C]."""
    rows: torch.Tensor
    b: int
    t: int
    h: int
    w: int

    @property
    def frames(self) -> int:
        return self.b * self.t

    @property
    def hw(self) -> int:
        return self.h * self.w

    @property
    def c(self) -> int:
        return self.rows.shape[1]

    def like(self, rows, h=None, w=None) -> "Act":
        return Act(rows, self.b, self.t, self.h if h is None else h, self.w if w is None else w)


class CfgShare:
    """Batched classifier-free guidance: the n passes of one sampler step (cond, uncond[, image-only]; reference
    ddim.py:226-233, ddim_multiplecond.py:226-236 run them as n UNet calls) get the SAME latent, concat conditioning,
    timestep and fps -- they differ only through the cross-attention context.  Everything in front of the first
    cross-attention (input conv, the initial temporal transformer, the first ResBlock + temporal convolutions, the first
    spatial self-attention) therefore computes n identical copies: it runs ONCE at batch b and is repeated where the
    paths part.  Exact (the identical copies were bit-identical to begin with: no cross-sample term anywhere).

    `branches=True` (TC_CFG_STREAMS=1, UNetModel._forward_branches): what follows the shared part is not one batch-(n b)
    pass but n batch-b passes, each on its own HIP stream -- the passes are independent from the first cross-attention
    on, so one pass's kernels fill the CUs another's leave idle (launch ramps and tails, the 64-block GroupNorms, the
    low-resolution levels).  Pass 0 RECORDS what it computes in front of the split (`cached`), passes 1.. REPLAY those
    tensors instead of computing them and wait for the event pass 0 recorded at the split."""

    def __init__(self, n: int, emb1: torch.Tensor, branches: bool = False):
        self.n, self.emb1, self.done, self.branches = n, emb1, False, branches
        self.embn = emb1 if branches else ops.repeat_rows(emb1, n)
        self.memo, self.replay, self.fork = {}, False, None

    def emb(self) -> torch.Tensor:
        return self.embn if self.done else self.emb1

    def expand(self, act: "Act") -> "Act":
        return Act(ops.repeat_rows(act.rows, self.n), act.b * self.n, act.t, act.h, act.w)

    # ---- branches mode
    def begin(self, k: int):
        """Start pass k: pass 0 records the shared part, the others replay it."""
        self.replay, self.done = k > 0, False

    def cached(self, key, fn):
        """Result of `fn()` for a piece of the network that may lie in front of the split.  Plain mode: just `fn()`.
        Branches mode: pass 0 keeps the result if the split has not happened by the time `fn` returns (a piece that
        CONTAINS the split is not kept: its inner pieces are); the other passes take a kept result instead of computing."""
        if not self.branches:
            return fn()
        if self.replay and key in self.memo:
            return self.memo[key]
        v = fn()
        if not self.replay and not self.done:
            self.memo[key] = v
        return v

    def split(self, x: torch.Tensor):
        """The point where the passes part (behind the first spatial self-attention).  Pass 0 marks it on its stream."""
        if self.branches and not self.replay and x.is_cuda:
            self.fork = torch.cuda.current_stream(x.device).record_event()
        self.done = True


class SourceKey:
    """Identity of the tensors a cache was filled from.  Holds STRONG references, so the caching
    allocator cannot recycle their storage under a new tensor while the cache lives, and compares by
    object identity + version counter -- never by `data_ptr()`, which a freed-and-reallocated tensor of
    the next clip can share (round-1 bug: clip 2 silently sampled with clip 1's conditioning)."""

    def __init__(self, tensors):
        self.tensors = list(tensors)
        self.versions = [None if t is None else t._version for t in self.tensors]

    def same(self, tensors) -> bool:
        tensors = list(tensors)
        if len(tensors) != len(self.tensors):
            return False
        for a, b, v in zip(self.tensors, tensors, self.versions):
            if a is not b or (a is not None and a._version != v):
                return False
        return True


# --------------------------------------------------------------------------- packers
def f32(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.float32).contiguous()


def pack_linear(w: torch.Tensor) -> torch.Tensor:
    """nn.Linear [N, K] / Conv1d-k1 [N, K, 1] / Conv2d-1x1 [N, K, 1, 1] -> bf16 [N, ceil8(K)]."""
    w = w.detach()
    w = w.reshape(w.shape[0], -1)
    n, k = w.shape
    kp = ceil_to(k, 8)
    out = torch.zeros((n, kp), dtype=BF16, device=w.device)
    out[:, :k] = w.to(BF16)
    return out


def fold_layernorm(w: torch.Tensor, b, gamma: torch.Tensor, beta: torch.Tensor):
    """nn.Linear(LayerNorm(x)) = ((x - mean) * rstd) @ (w * gamma)^T + (b + w @ beta): the affine half of the LayerNorm
    moved into the consumer's weight [N, K] and bias (fp32, exact in real arithmetic), so that the normalisation itself
    can run as the prologue of that GEMM (ops.gemm(..., a_norm_eps=...), csrc/gemm_ws.hip).  -> (w', b') fp32."""
    w = w.detach().reshape(w.shape[0], -1).to(torch.float32)
    bias = w @ beta.detach().to(torch.float32)
    if b is not None:
        bias = bias + b.detach().to(torch.float32)
    return w * gamma.detach().to(torch.float32)[None, :], bias.contiguous()


def pack_conv3x3(w: torch.Tensor, cin_pad: int | None = None) -> torch.Tensor:
    """Conv2d [Co, Ci, 3, 3] -> bf16 [Co, 9 * Cip], K index = (ky*3 + kx) * Cip + ci."""
    w = w.detach()
    co, ci, kh, kw = w.shape
    assert kh == 3 and kw == 3
    cip = ceil_to(ci, 64) if cin_pad is None else cin_pad
    out = torch.zeros((co, 9, cip), dtype=BF16, device=w.device)
    out[:, :, :ci] = w.permute(0, 2, 3, 1).reshape(co, 9, ci).to(BF16)
    return out.reshape(co, 9 * cip)


def pack_convt3(w: torch.Tensor) -> torch.Tensor:
    """Conv3d [Co, Ci, 3, 1, 1] -> bf16 [Co, 3 * Ci], K index = kt * Ci + ci."""
    w = w.detach()
    co, ci = w.shape[:2]
    assert tuple(w.shape[2:]) == (3, 1, 1) and ci % 64 == 0
    return w.reshape(co, ci, 3).permute(0, 2, 1).reshape(co, 3 * ci).to(BF16).contiguous()


def pack_geglu(w: torch.Tensor, b: torch.Tensor):
    """GEGLU proj [2F, C] (rows [0,F) values, [F,2F) gates) -> every 32 packed rows hold 16 value
    rows followed by their 16 gate rows (independent of the GEMM tile shape: the gate of packed
    column c sits at c + 16); bias likewise (fp32)."""
    w = w.detach()
    b = b.detach()
    f2, c = w.shape
    f = f2 // 2
    assert f % 16 == 0
    wv, wg = w[:f].reshape(f // 16, 16, c), w[f:].reshape(f // 16, 16, c)
    wp = torch.cat([wv, wg], dim=1).reshape(f2, c)
    bv, bg = b[:f].reshape(f // 16, 16), b[f:].reshape(f // 16, 16)
    bp = torch.cat([bv, bg], dim=1).reshape(f2)
    kp = ceil_to(c, 8)
    out = torch.zeros((f2, kp), dtype=BF16, device=w.device)
    out[:, :c] = wp.to(BF16)
    return out, bp.to(torch.float32).contiguous()


class PackedModule(nn.Module):
    """Base for modules that keep reference-layout nn.Parameters (so
    `load_state_dict(strict=True)` works with the reference's checkpoints) and a
    lazily built dict of packed device tensors used by forward."""

    # Bumped whenever ANY module drops its packed tensors (new weights, new device): captured hipGraphs have the
    # packed pointers baked in, so their owners compare this against the value at capture time (graph_epoch()).
    _epoch = [0]

    @staticmethod
    def graph_epoch() -> int:
        return PackedModule._epoch[0]

    def __init__(self):
        super().__init__()
        self._pk = None

    def _pack(self) -> dict:
        raise NotImplementedError

    @property
    def pk(self) -> dict:
        if self._pk is None:
            with torch.no_grad():
                self._pk = self._pack()
        return self._pk

    def invalidate(self):
        PackedModule._epoch[0] += 1
        for m in self.modules():
            if isinstance(m, PackedModule):
                m._pk = None

    def _apply(self, fn, recurse=True):           # .cuda() / .to(): packed copies are stale
        self._pk = None
        PackedModule._epoch[0] += 1
        return super()._apply(fn, recurse)

    def _load_from_state_dict(self, *a, **k):     # new weights: repack on next use
        self._pk = None
        PackedModule._epoch[0] += 1
        return super()._load_from_state_dict(*a, **k)
