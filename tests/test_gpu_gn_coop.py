"""The cooperative single-launch GroupNorm (csrc/gn_coop.hip, ABI 11: tc_groupnorm_coop) -- reference call sites
GroupNormSpecific / nn.GroupNorm (+ SiLU) of lvdm/basics.py:76-87, openaimodel3d.py:152-154,176-179,255-266 -- against the
fp32 statement of the operator (tests/emu_ops.py), against the three-launch path it replaces, and for the properties its
inter-block exchange must have: bit-identical reruns (the reduction order does not depend on which block arrives last), counters
left at zero, many rounds through the same blocks, eager == hipGraph replay."""
import pytest
import torch

from emu_ops import EmuOps
from test_gpu_gemm8 import env
from test_gpu_ops import check, rnd

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.fixture(scope="module")
def hip():
    from tooncrafter_amd.ops import HipOps
    return HipOps()


@pytest.fixture(scope="module")
def emu():
    return EmuOps(round_bf16=True)


def _inputs(samples, rows, c, seed=50):
    x = rnd(samples * rows, c, seed=seed) * 2.0 + 0.5
    g, b = rnd(c, seed=seed + 1, dtype=torch.float32) * 0.1 + 1.0, rnd(c, seed=seed + 2, dtype=torch.float32) * 0.1
    return x, g, b


def _coop(hip, x, g, b, samples, rows, silu=True, cap=None, eps=1e-5):
    """Through HipOps with TC_GN_COOP=2 (wherever a plan exists); fails if the library did not take the problem."""
    kw = {"TC_GN_COOP": 2}
    if cap is not None:
        kw["TC_GN_COOP_CAP"] = cap
    with env(**kw):
        assert hip.lib.tc_groupnorm_coop_grid(samples, rows, x.shape[1]) > 0, "cooperative kernel refused the problem"
        return hip.groupnorm(x, g, b, samples=samples, rows=rows, eps=eps, silu=silu)


SHAPES = [(32, 2560, 320), (2, 40960, 320), (32, 2560, 960), (2, 10240, 640), (32, 640, 1280), (2, 2560, 1280),
          (32, 40, 1280), (2, 640, 2560), (3, 1000, 128), (1, 4099, 512), (5, 37, 1920), (2, 1, 64), (16, 40960, 256)]


@pytest.mark.parametrize("samples,rows,c", SHAPES)
@pytest.mark.parametrize("silu", [True, False])
def test_coop_groupnorm_vs_fp32_and_three_launch(hip, emu, samples, rows, c, silu):
    x, g, b = _inputs(samples, rows, c)
    got = _coop(hip, x, g, b, samples, rows, silu)
    check(got, emu.groupnorm(x, g, b, samples=samples, rows=rows, eps=1e-5, silu=silu), f"coop groupnorm s{samples} r{rows} c{c} silu{silu}")
    with env(TC_GN_COOP=0):
        base = hip.groupnorm(x, g, b, samples=samples, rows=rows, eps=1e-5, silu=silu)
    # same formula (E[x^2] - mean^2 in fp64 over fp32 partial sums), other chunking: a bf16 ulp on a few elements at most
    d = (got.float() - base.float()).abs().max().item()
    assert d <= 2.0 ** -6 * max(base.float().abs().max().item(), 1.0), d
    again = _coop(hip, x, g, b, samples, rows, silu)
    assert torch.equal(got, again), "cooperative GroupNorm is not bit-reproducible"
    sync = hip._gn_sync(x.device)
    torch.cuda.synchronize()
    assert int(sync.abs().max()) == 0, "counters not returned to zero"


@pytest.mark.parametrize("cap", [7, 33, 100])
def test_coop_groupnorm_many_rounds(hip, emu, cap):
    """Few co-resident blocks allowed: the same blocks walk many rounds of whole samples (the 960-channel case at full size)."""
    for samples, rows, c in ((9, 300, 320), (32, 160, 1280), (4, 700, 640)):
        x, g, b = _inputs(samples, rows, c, seed=60)
        got = _coop(hip, x, g, b, samples, rows, True, cap=cap)
        check(got, emu.groupnorm(x, g, b, samples=samples, rows=rows, eps=1e-5, silu=True), f"coop groupnorm cap{cap} s{samples} r{rows} c{c}")
        full = _coop(hip, x, g, b, samples, rows, True)
        d = (got.float() - full.float()).abs().max().item()
        assert d <= 2.0 ** -6 * max(full.float().abs().max().item(), 1.0), d


def test_coop_groupnorm_large_mean(hip):
    """mean 60, spread 1: as test_groupnorm_large_mean_two_pass, for the cooperative kernel's accumulation."""
    for samples, rows, c in ((2, 2560, 320), (2, 10240, 640)):
        gen = torch.Generator().manual_seed(77)
        x = (torch.randn(samples * rows, c, generator=gen) + 60.0).to(torch.bfloat16).to(DEV)
        g, b = torch.ones(c, device=DEV), torch.zeros(c, device=DEV)
        got = _coop(hip, x, g, b, samples, rows, False)
        xd = x.double().reshape(samples, rows, 32, c // 32).permute(0, 2, 1, 3)
        mean = xd.mean(dim=(2, 3), keepdim=True)
        var = ((xd - mean) ** 2).mean(dim=(2, 3), keepdim=True)
        ref = ((xd - mean) / (var + 1e-5).sqrt()).permute(0, 2, 1, 3).reshape(samples * rows, c)
        err = float((got.double() - ref).norm() / ref.norm())
        print(f"coop groupnorm mean 60 +- 1, s{samples} r{rows} c{c}: rel-L2 vs float64 {err:.3e}")
        assert err < 6e-3, err


def test_coop_groupnorm_sample_locality(hip):
    """Changing one sample changes that sample's rows and nothing else (a block that read another sample's statistics, or a
    counter shared by two samples, would show)."""
    samples, rows, c = 8, 500, 320
    x, g, b = _inputs(samples, rows, c, seed=70)
    y0 = _coop(hip, x, g, b, samples, rows)
    x2 = x.clone()
    x2[3 * rows:4 * rows] *= 1.7
    y1 = _coop(hip, x2, g, b, samples, rows)
    same = torch.ones(samples * rows, dtype=torch.bool, device=DEV)
    same[3 * rows:4 * rows] = False
    assert torch.equal(y0[same], y1[same]) and not torch.equal(y0[~same], y1[~same])


def test_coop_groupnorm_back_to_back_and_graph_replay(hip):
    """200 launches in a row on one stream (the counters are reused launch after launch), then the same under hipGraph replay."""
    samples, rows, c = 32, 2560, 320
    x, g, b = _inputs(samples, rows, c, seed=80)
    with env(TC_GN_COOP=2):
        ref = hip.groupnorm(x, g, b, samples=samples, rows=rows, eps=1e-5, silu=True)
        for _ in range(200):
            y = hip.groupnorm(x, g, b, samples=samples, rows=rows, eps=1e-5, silu=True)
        assert torch.equal(y, ref)
        torch.cuda.synchronize()
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=st):
                yg = hip.groupnorm(x, g, b, samples=samples, rows=rows, eps=1e-5, silu=True)
            for _ in range(20):
                graph.replay()
        torch.cuda.current_stream().wait_stream(st)
        torch.cuda.synchronize()
        assert torch.equal(yg, ref)
        assert int(hip._gn_sync(x.device).abs().max()) == 0
