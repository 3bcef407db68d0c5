"""The persistent 8-wave GEMM (csrc/gemm8.hip) walks tiles by id: block b takes ids b, b + G, b + 2G, ... and
`g8_tile_of` turns an id into (tile_m, tile_n).  The map must be a BIJECTION on [0, tiles_m * tiles_n) -- a hole leaves
an output tile unwritten, a duplicate computes one twice -- for every grid, including M-tile counts that are not
multiples of the group height and totals that are not multiples of the 8 XCDs.  This is the host-side restatement of
that function (same integer arithmetic), checked exhaustively on small grids and on the grids of the model.
"""
import itertools

import pytest

GM = 4      # G8_GM


def tile_of(idx, tiles_m, tiles_n, total):
    x, j = idx & 7, idx >> 3
    q, r = total >> 3, total & 7
    lin = x * q + min(x, r) + j
    per_group = GM * tiles_n
    g, within = divmod(lin, per_group)
    gm = min(GM, tiles_m - g * GM)
    tn, rem = divmod(within, gm)
    return g * GM + rem, tn


@pytest.mark.parametrize("tiles_m,tiles_n", list(itertools.product(range(1, 23), range(1, 12)))
                         + [(320, 10), (80, 20), (20, 40), (5, 5), (640, 2), (2560, 1), (16, 16), (32, 32), (321, 3)])
def test_tile_map_is_a_bijection(tiles_m, tiles_n):
    total = tiles_m * tiles_n
    seen = set()
    for idx in range(total):
        tm, tn = tile_of(idx, tiles_m, tiles_n, total)
        assert 0 <= tm < tiles_m and 0 <= tn < tiles_n, (idx, tm, tn)
        seen.add((tm, tn))
    assert len(seen) == total


def test_an_xcd_walks_compact_patches():
    """32 consecutive tiles of one XCD (what its 32 CUs hold at a time) touch few distinct operand panels."""
    tiles_m, tiles_n = 32, 32
    total = tiles_m * tiles_n
    ids = [8 * j + 3 for j in range(32)]                       # XCD 3, first round of a 256-block grid
    tiles = [tile_of(i, tiles_m, tiles_n, total) for i in ids]
    assert len({t[0] for t in tiles}) + len({t[1] for t in tiles}) <= 12
