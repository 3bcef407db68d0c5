#!/usr/bin/env python3
"""Index-level model of the tap-reuse 3x3 convolution planned in docs/LAB_NOTEBOOK.md section 8 (item 1), in numpy.

It moves data exactly the way the kernel will -- 2-D output tiles of TH x TW pixels, one halo tile of
(TH+2) x (TW+2) pixels per 64-channel chunk brought into an LDS image by 1-KiB DMA pieces of 8 pixels
(16-byte chunks XOR-swizzled on the source side by (q>>1)&7 of the halo pixel index q), MFMA A fragments of
every tap read from that one image at a pixel offset, weights in (channel chunk, tap, channel) K order --
and checks (a) the result against a direct convolution and (b) that every ds_read_b128 lane group of every
tap's fragment read touches 16 distinct 16-byte bank slots (conflict-free), for both parities of the offset.
Design artefact for the next round: nothing in the product imports it."""
import numpy as np

TH, TW, CK = 8, 32, 64           # output tile 8 x 32 pixels = 256 GEMM rows; 64 channels per chunk
HW_H, HW_W = TH + 2, TW + 2      # halo
LANE_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
               list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]   # ds_read_b128 groups (lanes 0-31)


def lds_image(x_img, y0, x0, c0):
    """x_img: (H, W, C) one frame.  Returns the LDS image [n_pixels_padded, 8 chunks, 8 values] as the DMA fills it:
    piece p covers halo pixels 8p..8p+7; lane l writes physical chunk l&7 of pixel 8p + (l>>3) with the LOGICAL
    chunk (l&7) ^ ((q>>1)&7); pixels outside the image arrive as zeros (out-of-range buffer offsets)."""
    H, W, _ = x_img.shape
    nq = HW_H * HW_W
    img = np.zeros(((nq + 7) // 8 * 8, 8, 8), dtype=x_img.dtype)
    for q in range(nq):
        hy, hx = divmod(q, HW_W)
        iy, ix = y0 + hy - 1, x0 + hx - 1
        src = np.zeros(CK, dtype=x_img.dtype)
        if 0 <= iy < H and 0 <= ix < W:
            src = x_img[iy, ix, c0:c0 + CK]
        for phys in range(8):
            logical = phys ^ ((q >> 1) & 7)
            img[q, phys] = src[logical * 8:logical * 8 + 8]
    return img


def read_fragment(img, q_rows, kk, half):
    """A-operand fragment of v_mfma_f32_32x32x16_bf16: lane (r, half) holds 8 values k = kk*16 + half*8 + j of row r.
    q_rows: the 32 halo pixel indices of the fragment's rows.  Returns [32, 8] and the bank slots touched."""
    logical = kk * 2 + half
    out = np.zeros((32, 8), dtype=img.dtype)
    slots = []
    for r, q in enumerate(q_rows):
        phys = logical ^ ((q >> 1) & 7)
        out[r] = img[q, phys]
        slots.append((q * 8 + phys) % 16)            # 16-byte slot within the 256-byte bank row
    return out, slots


def conv_by_halo(x, w, frames, H, W):
    """x: (frames*H*W, C) channels-last rows; w: (N, 3, 3, C).  Returns (frames*H*W, N)."""
    C, N = x.shape[1], w.shape[0]
    out = np.zeros((frames * H * W, N), dtype=np.float64)
    worst = 0
    for f in range(frames):
        x_img = x[f * H * W:(f + 1) * H * W].reshape(H, W, C)
        for y0 in range(0, H, TH):
            for x0 in range(0, W, TW):
                acc = np.zeros((TH * TW, N), dtype=np.float64)
                for c0 in range(0, C, CK):
                    img = lds_image(x_img, y0, x0, c0)
                    for tap in range(9):
                        dy, dx = tap // 3 - 1, tap % 3 - 1
                        wt = w[:, tap // 3, tap % 3, c0:c0 + CK].astype(np.float64)        # (N, 64), K order (chunk, tap, ci)
                        for sub in range(TH * TW // 32):                                      # 32-row MFMA sub-tiles
                            py, px0 = divmod(sub * 32, TW)
                            q_rows = [(py + dy + 1) * HW_W + (px0 + j + dx + 1) for j in range(32)]
                            for kk in range(4):
                                for half in range(2):
                                    frag, slots = read_fragment(img, q_rows, kk, half)
                                    for grp in LANE_GROUPS:
                                        worst = max(worst, 16 - len({slots[l] for l in grp}))
                                    k0 = kk * 16 + half * 8
                                    acc[sub * 32:sub * 32 + 32] += frag.astype(np.float64) @ wt[:, k0:k0 + 8].T
                for p in range(TH * TW):
                    py, px = divmod(p, TW)
                    if y0 + py < H and x0 + px < W:
                        out[(f * H + y0 + py) * W + x0 + px] = acc[p]
    return out, worst


def conv_direct(x, w, frames, H, W):
    C, N = x.shape[1], w.shape[0]
    xi = np.zeros((frames, H + 2, W + 2, C), dtype=np.float64)
    xi[:, 1:-1, 1:-1] = x.reshape(frames, H, W, C)
    out = np.zeros((frames, H, W, N), dtype=np.float64)
    for ky in range(3):
        for kx in range(3):
            out += xi[:, ky:ky + H, kx:kx + W] @ w[:, ky, kx].astype(np.float64).T
    return out.reshape(frames * H * W, N)


def check(frames=2, H=10, W=40, C=128, N=24, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((frames * H * W, C)).astype(np.float32)
    w = rng.standard_normal((N, 3, 3, C)).astype(np.float32)
    got, worst = conv_by_halo(x, w, frames, H, W)
    ref = conv_direct(x, w, frames, H, W)
    err = np.abs(got - ref).max() / np.abs(ref).max()
    return err, worst


if __name__ == "__main__":
    err, worst = check()
    print(f"tap-reuse model vs direct 3x3 convolution: max rel err {err:.2e}; worst ds_read_b128 lane-group conflict "
          f"{worst} extra slots (0 = conflict-free)")
