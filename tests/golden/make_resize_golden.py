#!/usr/bin/env python3
"""Writes tests/golden/resize_kornia.npz: the pin of `tooncrafter_amd.lvdm.condition.kornia_resize` (the
last leg that had no independent check: VERDICT r5 missing #4).

The reference preprocesses the conditioning image with `kornia.geometry.resize(x, (224, 224), interpolation='bicubic',
align_corners=True, antialias=True)` (reference lvdm/modules/encoders/condition.py:322-326).  kornia is a third-party
package absent from /root/reference and from this image (unpinned in the reference's requirements.txt), so its PUBLISHED
algorithm is restated here a second time, with NO code shared with the product's restatement and on different libraries:

  * antialias (kornia/geometry/transform/affwarp.py `resize`): only when downscaling (max factor > 1); per axis
    sigma = max((factor - 1) / 2, 0.001) -- the skimage rule kornia cites -- kernel size int(max(4 sigma, 3)) made odd;
    `gaussian_blur2d(input, ks, sigmas)` = separable correlation with exp(-x^2 / 2 sigma^2) / sum over the ks taps,
    border 'reflect' (the edge sample is NOT repeated).
    Here: `scipy.ndimage.gaussian_filter1d(..., sigma, radius=ks // 2, mode='mirror')` in float64 -- scipy builds its own
    truncated, normalised Gaussian; its 'mirror' is torch's / kornia's 'reflect';
  * then `torch.nn.functional.interpolate(mode='bicubic', align_corners=True)` WITHOUT torch's antialias: cubic convolution
    (Keys, A = -0.75) on source coordinate dst * (in - 1) / (out - 1), the four taps' indices clamped to the image.
    Here: a numpy float64 gather of the four taps per axis.

Inputs are seeded; the file holds inputs and expected outputs only.  Run anywhere (CPU):
    python tests/golden/make_resize_golden.py
"""
import os

import numpy as np
from scipy import ndimage

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "resize_kornia.npz")

# (name, (C, H, W), (out_h, out_w)): the BASELINE frame, a square crop, one axis up / one down, pure upscaling (no blur),
# a strong downscale (factor 3.57: a 5-tap kernel), and the identity size
# (channels are independent in every step, so one or two per case keep the fixture small)
CASES = [("frame_320x512", (2, 320, 512), (224, 224)),
         ("square_256", (1, 256, 256), (224, 224)),
         ("mixed_200x600", (1, 200, 600), (224, 224)),
         ("up_100x150", (2, 100, 150), (224, 224)),
         ("down_800x300", (1, 800, 300), (224, 224)),
         ("same_224", (1, 224, 224), (224, 224))]


def blur_rule(factor):
    """kornia's (sigma, kernel size) for one axis"""
    sigma = max((factor - 1.0) / 2.0, 0.001)
    ks = int(max(2.0 * 2 * sigma, 3))
    return sigma, ks + 1 if ks % 2 == 0 else ks


def cubic_taps(t, a=-0.75):
    """Keys cubic convolution weights of the taps at offsets -1, 0, 1, 2 for fractional position t in [0, 1)"""
    def near(x):    # |x| <= 1
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0

    def far(x):     # 1 < |x| < 2
        return ((a * x - 5.0 * a) * x + 8.0 * a) * x - 4.0 * a
    return np.stack([far(t + 1.0), near(t), near(1.0 - t), far(2.0 - t)], -1)


def bicubic_axis(x, out, axis):
    n = x.shape[axis]
    src = np.arange(out, dtype=np.float64) * ((n - 1) / (out - 1) if out > 1 else 0.0)      # align_corners=True
    i0 = np.floor(src).astype(np.int64)
    w = cubic_taps(src - i0)                                                                # (out, 4)
    x = np.moveaxis(x, axis, -1)
    acc = 0.0
    for k in range(4):
        acc = acc + x[..., np.clip(i0 - 1 + k, 0, n - 1)] * w[:, k]
    return np.moveaxis(acc, -1, axis)


def expected(x, size):
    x = x.astype(np.float64)
    h, w = x.shape[-2:]
    factors = (h / size[0], w / size[1])
    meta = []
    if (h, w) == tuple(size):                 # kornia returns the input untouched
        return x, [(0.0, 0), (0.0, 0)]
    if max(factors) > 1:
        for axis, f in zip((-2, -1), factors):
            sigma, ks = blur_rule(f)
            meta.append((sigma, ks))
            x = ndimage.gaussian_filter1d(x, sigma, axis=axis, radius=ks // 2, mode="mirror")
    else:
        meta = [(0.0, 0), (0.0, 0)]
    x = bicubic_axis(x, size[0], -2)
    x = bicubic_axis(x, size[1], -1)
    return x, meta


def main():
    out = {}
    for i, (name, shape, size) in enumerate(CASES):
        rng = np.random.default_rng(4100 + i)
        # images in [-1, 1] like the script's frames: a smooth field + texture, so that blur and interpolation both matter
        c, h, w = shape
        yy, xx = np.meshgrid(np.linspace(0, 1, h), np.linspace(0, 1, w), indexing="ij")
        base = np.stack([np.sin(6.0 * (k + 1) * xx + 2.0 * yy) * np.cos(5.0 * yy * (k + 1)) for k in range(c)])
        x = np.clip(0.6 * base + 0.4 * rng.standard_normal(shape), -1.0, 1.0).astype(np.float16)   # stored as fp16: exact in fp32
        y, meta = expected(x.astype(np.float32), size)
        out[f"{name}.x"] = x
        out[f"{name}.y"] = y.astype(np.float32)
        out[f"{name}.size"] = np.array(size)
        out[f"{name}.blur"] = np.array(meta, dtype=np.float64)          # [(sigma_y, ks_y), (sigma_x, ks_x)]
        print(f"{name}: {shape} -> {size}, blur (sigma, taps) rows {meta[0]}, columns {meta[1]}")
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
