"""Output path of the interpolation scripts: decoded clip -> uint8 frames -> file.

Mirror of reference scripts/evaluation/inference.py:135-155 (`save_results_seperate`): same
arguments, same file naming, same arithmetic -- but clamp / scale / uint8 / (c t h w)->(t h w c)
run on the device in one pass (tc_video_to_u8) BEFORE anything leaves the GPU, so the host copy
and the multi-GPU gather move one byte per sample instead of four.

h264 encoding is not part of the accelerated path: the frames go to `torchvision.io.write_video`
when torchvision is installed (as in the reference), to a caller-supplied `writer`, or -- on images
without a video encoder, like the build container -- through the self-contained writer of `mp4.py`:
the same container and codec (one avc1 track, H.264), every macroblock coded as I_PCM, i.e. lossless
after the yuv420p conversion where libx264 at crf 10 is not.
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional

import numpy as np
import torch

from . import dist as tdist
from . import ops


def clip_to_uint8(samples: torch.Tensor, loop: bool = False) -> torch.Tensor:
    """(b, 3, t, h, w) fp32 -> (b, t, h, w, 3) uint8 on the device of `samples`
    (inference.py:146-153; `loop` drops the last frame, :146-147)."""
    if loop:
        samples = samples[:, :, :-1]
    return ops.video_to_uint8(samples.to(torch.float32))


def default_writer(path: str, frames: torch.Tensor, fps: int) -> str:
    """frames: (t, h, w, 3) uint8 CPU.  Returns the path actually written."""
    try:
        import torchvision                                    # noqa: F401
        if getattr(torchvision, "__tooncrafter_shim__", False):
            raise ImportError("dropin's torchvision shim has no video encoder")
    except Exception:
        from .mp4 import write_mp4
        t, h, w, _ = frames.shape
        if h % 2 or w % 2:                                    # yuv420p needs even sizes: keep the raw frames instead
            alt = os.path.splitext(path)[0] + ".npy"
            np.save(alt, frames.numpy())
            return alt
        return write_mp4(path, frames, fps)
    torchvision.io.write_video(path, frames, fps=fps, video_codec='h264', options={'crf': '10'})
    return path


def save_results_seperate(prompt, samples, filename, fakedir, fps=10, loop=False,
                          writer: Optional[Callable[[str, torch.Tensor, int], str]] = None) -> List[str]:
    """Reference signature (inference.py:135) plus an optional `writer(path, frames_thwc_u8, fps)`."""
    prompt = prompt[0] if isinstance(prompt, list) else prompt
    if samples is None:
        return []
    writer = writer or default_writer
    frames = clip_to_uint8(samples.detach(), loop=loop).cpu()
    outdir = fakedir.replace('samples', 'samples_separate')
    os.makedirs(outdir, exist_ok=True)
    written = []
    for i in range(frames.shape[0]):
        path = os.path.join(outdir, f'{filename.split(".")[0]}_sample{i}.mp4')
        written.append(writer(path, frames[i], fps))
    return written


def gather_frames(samples: torch.Tensor, dst: int = 0, loop: bool = False):
    """Multi-GPU writer side: convert on every rank, gather the uint8 frames to `dst` (one RCCL gather,
    7.9 MB per 16-frame clip instead of 31.5 MB fp32).  Returns the per-rank list on `dst`, None elsewhere."""
    return tdist.gather_clips(clip_to_uint8(samples, loop=loop), dst=dst)
