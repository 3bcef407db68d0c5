"""CPU mirror of csrc/conv_halo.hip (the tap-reuse 3x3 / temporal convolution, TC_CONV_HALO, off by default).

The kernel was written in a session without GPU access.  What CAN be checked here is its index arithmetic: this file
restates every address formula of the kernel -- block -> patch, the per-thread halo vectors (source offset, validity,
swizzled LDS address), the W-tile DMA image, the MFMA fragment addresses of every tap, the 16x16x32 operand / result
lane layouts and the epilogue's tile row -> output row map -- with the kernel's own variable names, moves numbers
through a byte-addressed LDS model exactly as the lanes would, and compares the result with a direct convolution.
It also counts LDS bank conflicts of every fragment read and halo write with the lane groups of
/opt/skills/guides/MI355X_MICROARCH.md (LDS table).  It does not (cannot) check waits, barriers or the compiler."""
import numpy as np
import pytest

CH_HX, CH_BN, CH_WT, CH_NT, TC_BK, CH_A_BYTES, CH_W_STAGE = 18, 160, 80, 5, 64, 28 * 1024, 160 * 128
CONV3x3, CONVT3 = 1, 2

# ds_read_b128: four groups of 16 lanes, one LDS cycle each when their 16-byte bank slots ((addr / 16) % 16) differ
READ_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
               list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
               list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
               list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]


def read_conflicts(addrs):
    """addrs: 64 byte addresses of one ds_read_b128.  Extra LDS cycles (0 = conflict-free)."""
    extra = 0
    for g in READ_GROUPS:
        slots = {}
        for l in g:
            slots.setdefault((addrs[l] // 16) % 16, set()).add(addrs[l])
        extra += max(len(v) for v in slots.values()) - 1
    return extra


def write_conflicts(addrs):
    """ds_write_b128: 8 groups of 8 contiguous lanes, bank (a / 4) % 32 over the 4 dwords each lane stores."""
    extra = 0
    for g0 in range(0, 64, 8):
        banks = {}
        for l in range(g0, g0 + 8):
            if addrs[l] is None:
                continue
            for d in range(4):
                banks.setdefault(((addrs[l] // 4) + d) % 32, set()).add(addrs[l] + 4 * d)
        if banks:
            extra += max(len(v) for v in banks.values()) - 1
    return extra


class HaloKernelModel:
    """One block of conv_halo_kernel<GATHER>, lane by lane."""

    def __init__(self, gather, x, w, frames, h, wd, cin, n, lda=None, WM=2, KS=1):
        self.g, self.x, self.w = gather, x, w
        self.frames, self.h, self.wd, self.cin, self.n = frames, h, wd, cin, n
        self.lda = lda or cin
        self.WM, self.PY, self.threads = WM, 5 * WM, 128 * WM          # WM = 4: the tall 320-row patch on eight waves
        self.KS = KS                                                    # KS = 2: two 4-wave groups, each with half of the channel chunks
        assert KS == 1 or (WM == 2 and (cin // TC_BK) % 2 == 0)
        self.taps = 9 if gather == CONV3x3 else 3
        self.hy = self.PY + 2 if gather == CONV3x3 else self.PY
        self.npix = self.hy * CH_HX
        self.a_bytes = (self.npix * 128 + 1023) // 1024 * 1024
        self.hw = h * wd
        self.m = frames * self.hw
        self.read_extra = self.write_extra = 0

    def tiles(self):
        if self.g == CONV3x3:
            per_img = (self.h // self.PY) * (self.wd // 16)
            return self.frames * per_img, per_img
        per_img = self.hw // self.PY
        return (self.frames // 16) * per_img, per_img

    def run_block(self, tile_m, tile_n, out):
        tiles_m, per_img = self.tiles()
        img, pin = divmod(tile_m, per_img)
        Y0 = X0 = 0
        if self.g == CONV3x3:
            tpx = self.wd // 16
            ty0 = pin // tpx
            Y0, X0 = ty0 * self.PY, (pin - ty0 * tpx) * 16
            m00 = (img * self.h + Y0) * self.wd + X0
            ys, xs = self.wd, 1
            row_lo = max(m00 - self.wd - 1, 0)
        else:
            m00 = img * 16 * self.hw + pin * self.PY
            ys, xs = 1, self.hw
            row_lo = m00
        sA = {}          # 16-byte unit index -> 8 values
        sW = [dict(), dict()]
        # per-thread halo vectors
        hv = []
        assert self.npix * 8 <= 7 * self.threads
        for tid in range(self.threads):
            mine = []
            for i in range(7):
                v = tid + self.threads * i
                q, seg = v >> 3, v & 7
                if self.g == CONV3x3:
                    hy, hx = divmod(q, CH_HX)
                else:
                    hx, hy = divmod(q, self.PY)             # pixel-fastest: a frame's pixels are adjacent source rows
                pix = hy * CH_HX + hx if q < self.npix else self.npix
                ok = q < self.npix
                if self.g == CONV3x3:
                    iy, ix = Y0 + hy - 1, X0 + hx - 1
                    ok = ok and 0 <= iy < self.h and 0 <= ix < self.wd
                    src = (img * self.h + iy) * self.wd + ix
                else:
                    ok = ok and 1 <= hx <= 16
                    src = (img * 16 + (hx - 1)) * self.hw + pin * self.PY + hy
                off = ((src - row_lo) * self.lda * 2 + seg * 16) if ok else None
                if ok:
                    assert 0 <= off < 2 ** 31 and src < self.m
                lds = pix * 128 + ((seg ^ (pix & 7)) << 4) if pix < self.npix else None
                mine.append((off, lds))
            hv.append(mine)

        def fill_halo(c):
            for wave in range(2 * self.WM):
                for i in range(7):
                    addrs = [hv[wave * 64 + l][i][1] for l in range(64)]
                    self.write_extra += write_conflicts(addrs)
            for tid in range(self.threads):
                for off, lds in hv[tid]:
                    if lds is None:
                        continue
                    assert lds % 16 == 0 and lds + 16 <= self.a_bytes
                    if off is None:
                        sA[lds // 16] = np.zeros(8)                      # outside the image: zero
                    else:
                        byte = row_lo * self.lda * 2 + off + c * 128
                        row, col = divmod(byte // 2, self.lda)
                        val = self.x[row, col:col + 8].astype(np.float64)
                        sA[lds // 16] = val

        RSTEP = 16 * self.WM
        RB = (CH_BN + RSTEP - 1) // RSTEP
        PIECE = RSTEP * 128

        def request_w(k0, stage):
            sW[stage].clear()                                  # a stale unit read later would be a bug of the model's subject
            for tid in range(self.threads):
                lrow = tid >> 3
                wchunk = (tid & 7) ^ ((lrow >> 1) & 7)
                wave, lane = tid >> 6, tid & 63
                for i in range(RB):
                    if not (self.WM == 2 or i < RB - 1 or wave < 4):
                        continue
                    nl = lrow + RSTEP * i
                    assert nl < CH_BN                              # (the kernel would request zeros here; no such lane remains)
                    nn = tile_n * CH_BN + nl
                    dst = wave * 1024 + i * PIECE + lane * 16
                    assert dst + 16 <= CH_W_STAGE and dst == nl * 128 + (tid & 7) * 16
                    sW[stage][dst // 16] = self.w[nn, k0 + wchunk * 8:k0 + wchunk * 8 + 8].astype(np.float64)

        acc = np.zeros((2 * self.WM, CH_NT, CH_NT, 64, 4))     # wave, i, j, lane, reg

        def compute(stage, shift):
            for wave in range(2 * self.WM):
                wm, wn = wave >> 1, wave & 1
                lanes = np.arange(64)
                frow, fq = lanes & 15, lanes >> 4
                for ks in range(2):
                    af, bf = [], []
                    for i in range(CH_NT):
                        hp = (wm * 5 + i) * CH_HX + frow + shift
                        assert hp.max() < self.npix
                        a_addr = ((hp << 7) + ((fq ^ (hp & 7)) << 4)) ^ (ks << 6)
                        self.read_extra += read_conflicts(list(a_addr))
                        af.append(np.stack([sA[a // 16] for a in a_addr]))          # [lane, 8]
                    b_sw = ((wn * CH_WT + frow) >> 1) & 7
                    cb = ((ks * 4 + fq) ^ b_sw) << 4
                    for j in range(CH_NT):
                        b_addr = (wn * CH_WT + j * 16 + frow) * 128 + cb
                        self.read_extra += read_conflicts(list(b_addr))
                        bf.append(np.stack([sW[stage][a // 16] for a in b_addr]))
                    for i in range(CH_NT):
                        # operand matrices of v_mfma_f32_16x16x32_bf16: lane l holds row l & 15, k = 8 (l >> 4) .. +7
                        A = np.zeros((16, 32))
                        for l in range(64):
                            A[l & 15, 8 * (l >> 4):8 * (l >> 4) + 8] = af[i][l]
                        for j in range(CH_NT):
                            B = np.zeros((16, 32))
                            for l in range(64):
                                B[l & 15, 8 * (l >> 4):8 * (l >> 4) + 8] = bf[j][l]
                            D = A @ B.T                                                # [row, col]
                            for l in range(64):                                        # result: col = l & 15, row = 4 (l >> 4) + r
                                for r in range(4):
                                    acc[wave, i, j, l, r] += D[4 * (l >> 4) + r, l & 15]

        nch = (self.cin // TC_BK) // self.KS
        total = np.zeros_like(acc)
        for grp in range(self.KS):                 # the groups own separate LDS buffers: run one after the other on fresh state
            sA.clear(); sW[0].clear(); sW[1].clear(); acc[...] = 0.0
            c0 = grp * nch
            nk = self.taps * nch
            request_w(c0 * TC_BK, 0)
            fill_halo(c0)
            c = tap = ty = tx = 0
            for kb in range(nk):
                st = kb & 1
                ntap, nc, nty, ntx = tap + 1, c, ty, tx + 1
                if ntx == 3:
                    ntx, nty = 0, ty + 1
                if ntap == self.taps:
                    ntap, nc, nty, ntx = 0, c + 1, 0, 0
                more = kb + 1 < nk
                refill = more and ntap == 0
                if more:
                    request_w(ntap * self.cin + (c0 + nc) * TC_BK, st ^ 1)
                compute(st, ty * CH_HX + tx if self.g == CONV3x3 else tx)
                if refill:
                    fill_halo(c0 + nc)
                tap, c, ty, tx = ntap, nc, nty, ntx
            assert c == nch and tap == 0            # the counters have stepped past the last (chunk, tap)
            total += acc                            # group 1 hands its accumulators to group 0: acc0 + acc1
        acc = total
        # epilogue map
        for wave in range(2 * self.WM):
            wm, wn = wave >> 1, wave & 1
            col_w0 = tile_n * CH_BN + wn * CH_WT
            for i in range(CH_NT):
                row_base = m00 + (wm * 5 + i) * ys
                for j in range(CH_NT):
                    for l in range(64):
                        frow, fq = l & 15, l >> 4
                        for r in range(4):
                            lr, col = fq * 4 + r, j * 16 + frow
                            out[row_base + lr * xs, col_w0 + col] = acc[wave, i, j, l, r]


def direct_conv(gather, x, w, frames, h, wd, cin, n):
    m = frames * h * wd
    out = np.zeros((m, n))
    xf, wf = x.astype(np.float64), w.astype(np.float64)
    if gather == CONV3x3:
        xi = xf[:, :cin].reshape(frames, h, wd, cin)
        o = out.reshape(frames, h, wd, n)
        for t in range(9):
            dy, dx = t // 3 - 1, t % 3 - 1
            wt = wf[:, t * cin:(t + 1) * cin]
            for y in range(h):
                for xx in range(wd):
                    iy, ix = y + dy, xx + dx
                    if 0 <= iy < h and 0 <= ix < wd:
                        o[:, y, xx] += xi[:, iy, ix] @ wt.T
    else:
        T = 16
        xi = xf[:, :cin].reshape(frames // T, T, h * wd, cin)
        o = out.reshape(frames // T, T, h * wd, n)
        for t in range(3):
            wt = wf[:, t * cin:(t + 1) * cin]
            for f in range(T):
                sf = f + t - 1
                if 0 <= sf < T:
                    o[:, f] += xi[:, sf] @ wt.T
    return out


@pytest.mark.parametrize("gather,frames,h,wd,cin,n,lda,WM,KS", [
    (CONV3x3, 2, 20, 32, 128, 160, None, 2, 1),    # 2 x 2 patches per frame: every border kind; two channel chunks (one refill)
    (CONV3x3, 1, 10, 16, 64, 320, 192, 2, 1),      # one patch = the whole image, two column tiles, strided rows
    (CONVT3, 16, 4, 5, 128, 160, None, 2, 1),      # two pixel patches per clip (hw = 20), one clip
    (CONVT3, 32, 2, 5, 64, 160, 128, 2, 1),        # two clips, strided rows
    (CONV3x3, 1, 40, 32, 128, 160, None, 4, 1),    # tall patches (20 rows, eight waves): 2 x 2 per frame
    (CONVT3, 16, 8, 5, 64, 160, 96, 4, 1),         # tall temporal patches: 20 pixels x 16 frames, two per clip
    (CONV3x3, 1, 10, 32, 256, 160, None, 2, 2),    # K split over two groups: chunks {0, 1} | {2, 3}, a refill in each
    (CONVT3, 16, 2, 5, 128, 160, 160, 2, 2),       # ... temporal: one chunk per group
])
def test_conv_halo_index_model_matches_direct_convolution(gather, frames, h, wd, cin, n, lda, WM, KS):
    rng = np.random.default_rng(5)
    lda = lda or cin
    m = frames * h * wd
    taps = 9 if gather == CONV3x3 else 3
    x = rng.integers(-3, 4, size=(m, lda)).astype(np.float32)     # small integers: every sum is exact
    w = rng.integers(-2, 3, size=(n, taps * cin)).astype(np.float32)
    model = HaloKernelModel(gather, x, w, frames, h, wd, cin, n, lda=lda, WM=WM, KS=KS)
    tiles_m, _ = model.tiles()
    assert tiles_m * 80 * WM == m
    out = np.full((m, n), np.nan)
    for tm in range(tiles_m):
        for tn in range(n // CH_BN):
            model.run_block(tm, tn, out)
    ref = direct_conv(gather, x, w, frames, h, wd, cin, n)
    assert not np.isnan(out).any(), "an output element was never written"
    assert np.array_equal(out, ref)
    assert model.read_extra == 0, f"{model.read_extra} extra LDS cycles in the fragment reads"
    assert model.write_extra == 0, f"{model.write_extra} extra LDS cycles in the halo writes"


def test_fragment_swizzle_choices():
    """hp & 7 is conflict-free for EVERY base of the 16 consecutive halo pixels a fragment read covers; gemm16's
    (row >> 1) & 7 is conflict-free for 16-aligned bases only -- which is all gemm16 needs, and not enough here."""
    def worst(f):
        bad = 0
        for base in range(0, 216 - 15):
            for ks in range(2):
                addrs = []
                for l in range(64):
                    hp, seg = base + (l & 15), ks * 4 + (l >> 4)
                    addrs.append(hp * 128 + ((seg ^ f(hp)) << 4))
                bad += read_conflicts(addrs) > 0
        return bad
    assert worst(lambda hp: hp & 7) == 0
    assert worst(lambda hp: (hp >> 1) & 7) > 0
    for base in range(0, 208, 16):
        addrs = [(base + (l & 15)) * 128 + (((l >> 4) ^ (((base + (l & 15)) >> 1) & 7)) << 4) for l in range(64)]
        assert read_conflicts(addrs) == 0


