"""Host arithmetic of the cooperative GroupNorm (csrc/gn_coop.hip, ABI 11): the decomposition `tc_groupnorm_coop_plan` hands
the kernel -- chunks per sample, samples per round, vectors per thread -- for every GroupNorm shape of the UNet and the decoder
and a sweep of ragged ones, against the invariants the kernel relies on (it has no bounds checks of its own beyond the chunk's
buffer extent, and a grid above the number of co-resident blocks would deadlock its spin-wait).  No GPU: plain C on the host."""
import ctypes as C
import itertools

import pytest

from tooncrafter_amd import _lib

UNET = [(s, r, c) for (s, r) in ((32, 2560), (2, 40960)) for c in (320, 640, 960)] + \
       [(s, r, c) for (s, r) in ((32, 640), (2, 10240)) for c in (640, 960, 1280, 1920)] + \
       [(s, r, c) for (s, r) in ((32, 160), (2, 2560)) for c in (1280, 1920, 2560)] + \
       [(s, r, c) for (s, r) in ((32, 40), (2, 640)) for c in (1280, 2560)]
DECODER = [(16, 163840, 128), (1, 2621440, 128), (16, 163840, 256), (16, 40960, 256), (16, 40960, 512), (16, 10240, 512),
           (16, 2560, 512), (14, 163840, 128)]


def plan(samples, rows, c, cap):
    lib = _lib.load()
    out = (C.c_int32 * 6)()
    ok = lib.tc_groupnorm_coop_plan(samples, rows, c, cap, out)
    return tuple(out) if ok else None


@pytest.mark.parametrize("cap", [512, 256, 96, 7])
def test_plan_invariants(cap):
    shapes = UNET + DECODER + [(s, r, c) for s, r, c in itertools.product((1, 3, 33), (1, 5, 37, 1000, 4099), (32, 128, 320, 4096))]
    taken = 0
    for samples, rows, c in shapes:
        p = plan(samples, rows, c, cap)
        if p is None:
            continue
        taken += 1
        nch, chunk_rows, spr, rounds, nv, grid = p
        rows_pp = 512 // (c // 8)
        assert rows_pp >= 1 and nv in (4, 8, 16)
        assert grid == spr * nch and 1 <= grid <= cap, (samples, rows, c, p)         # all blocks of a round co-resident
        assert spr * rounds >= samples and spr * (rounds - 1) < samples              # every sample in exactly one round
        assert nch * chunk_rows >= rows and (nch - 1) * chunk_rows < rows            # chunks tile the rows, none empty
        assert chunk_rows <= nv * rows_pp                                            # a chunk fits the block's registers
        assert nch <= 1024                                                           # the workspace layout's chunks per sample
        assert (chunk_rows * c * 2) < 2 ** 31                                        # the chunk's buffer extent is 31-bit
    assert taken > len(shapes) // 3


def test_plan_refuses_what_the_kernel_cannot_hold():
    assert plan(1, 10 ** 6, 320, 512) is None            # one sample of 640 MB: more chunks than co-resident blocks
    assert plan(2, 100, 48, 512) is None and plan(2, 100, 8192, 512) is None        # c % 32, c > 4096
    assert plan(0, 100, 320, 512) is None and plan(2, 0, 320, 512) is None
    assert plan(70000, 10, 320, 512) is None             # sample index must fit the counter buffer


def test_level0_runs_as_one_round_and_wide_concats_as_more():
    """UNet level 0 at C = 320 (52 MB) fits the register files in one round; the 960-channel concatenation (157 MB) runs as
    whole-sample rounds through the same blocks."""
    nch, chunk_rows, spr, rounds, nv, grid = plan(32, 2560, 320, 512)
    assert rounds == 1 and spr == 32 and grid <= 512 and nv == 16
    p = plan(32, 2560, 960, 512)
    assert p is not None and p[3] >= 3 and p[2] * p[3] >= 32
    # clip-wide statistics: two samples, hundreds of chunks each, one round
    p = plan(2, 40960, 320, 512)
    assert p[3] == 1 and p[2] == 2 and p[0] * 2 == p[5] <= 512


def test_rows_of_a_chunk_are_covered_exactly_once():
    """The kernel's thread -> row map: thread (rlane, col) of a block takes rows r0 + rlane + i * rows_pp, i < nv, below r1."""
    for samples, rows, c, cap in ((3, 1000, 320, 64), (2, 37, 1280, 512), (1, 4099, 128, 512)):
        nch, chunk_rows, spr, rounds, nv, grid = plan(samples, rows, c, cap)
        rows_pp = 512 // (c // 8)
        seen = [0] * rows
        for chunk in range(nch):
            r0, r1 = chunk * chunk_rows, min(rows, (chunk + 1) * chunk_rows)
            for rlane in range(rows_pp):
                nrow = r1 - r0 - rlane
                for i in range(nv):
                    if i * rows_pp < nrow:
                        seen[r0 + rlane + i * rows_pp] += 1
        assert seen == [1] * rows, (samples, rows, c)
