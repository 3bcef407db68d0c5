"""GroupNorm statistics from the producing GEMM (ABI 9: TcGemmParams.gn_part, tc_groupnorm_part; reference
lvdm/basics.py:76-87 normalising what lvdm/modules/networks/openaimodel3d.py:154,179,255-266 just computed): the HOST half
-- ResBlock / TemporalConvBlock passing the partial sums from each convolution to the norm behind it -- on the CPU
emulation, which restates the partial-sum contract (per block of rows and per column: sum and sum of squares of the
rounded outputs; the norm reduces them in fp64)."""
import torch

from emu_ops import EmuOps
from tooncrafter_amd import ops
from tooncrafter_amd.lvdm.common import Act
from tooncrafter_amd.lvdm.openaimodel3d import ResBlock


def _resblock():
    torch.manual_seed(0)
    blk = ResBlock(64, 128, 0.0, out_channels=64, use_temporal_conv=True).eval()
    with torch.no_grad():
        for p in blk.parameters():
            p.normal_(0, 0.05)
        for m in blk.modules():
            if isinstance(m, torch.nn.GroupNorm):
                m.weight.add_(1.0)
    blk.emb_slice = (0, 64)
    return blk


def _run(gn_rows):
    blk = _resblock()
    emu = EmuOps(round_bf16=True, gn_rows=gn_rows)
    prev = ops.set_backend(emu)
    try:
        b, t, h, w = 2, 4, 4, 8                                  # 32 rows per frame, 128 per clip
        g = torch.Generator().manual_seed(1)
        x = (torch.randn(b * t * h * w, 64, generator=g) + 0.5).to(torch.bfloat16)
        emb = torch.randn(b, 64, generator=g)
        with torch.no_grad():
            y = blk(Act(x, b, t, h, w), emb).rows.float()
    finally:
        ops.set_backend(prev)
    return y, emu


def test_resblock_with_producer_statistics_matches_the_two_pass_norm():
    y0, e0 = _run(0)
    y1, e1 = _run(16)                   # 16 | 32 rows per frame and | 128 rows per clip: all five norms take the partial sums
    assert e0.gn_part_made == 0 and e0.gn_part_used == 0
    assert e1.gn_part_made == 5 and e1.gn_part_used == 5        # conv1 -> GN2, conv2 -> tGN1, tconv1..3 -> tGN2..4
    rel = float((y1 - y0).norm() / y0.norm())
    print("ResBlock with / without producer statistics: rel-L2", rel)
    assert rel < 5e-3                   # E[x^2] - mean^2 in fp64 from fp32 block sums vs F.group_norm: bf16 rounding flips only


def test_blocks_that_straddle_samples_fall_back():
    """48 does not divide the 32 rows of a frame: the per-frame norm computes its own statistics; 128 rows per clip are
    not a multiple either -- every norm falls back, the result is the plain one bit for bit."""
    y0, _ = _run(0)
    y1, e1 = _run(48)
    assert e1.gn_part_made == 5 and e1.gn_part_used == 0
    assert torch.equal(y0, y1)
