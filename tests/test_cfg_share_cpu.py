"""Batched classifier-free guidance with the shared prefix (lvdm/common.py: CfgShare): the n guided passes of a sampler
step have the same latent, concat conditioning, timestep and fps, so the UNet runs what precedes its first cross-attention
once and repeats the rows there.  `replicas=n` on single-copy inputs must equal the plain batch-(n b) call on repeated
inputs -- the reference's n separate `apply_model` calls (ddim.py:226-233) -- and the sampler must only take the short cut
when the passes really share their concat conditioning."""
import pytest
import torch

from conftest import TINY_UNET_CFG, sub_state_dict
from emu_ops import EmuOps
from tooncrafter_amd import ops, synth
from tooncrafter_amd.lvdm.openaimodel3d import UNetModel


@pytest.fixture()
def emu_backend():
    # fp32 emulation WITHOUT the bf16 rounding of every operator result: the statement is about the arithmetic being the
    # same, and on the tiny net a rounding flipped by torch's batch-dependent summation order shows up at the 1e-2 level
    old = ops.set_backend(EmuOps(round_bf16=False))
    yield
    ops.set_backend(old)


@pytest.mark.parametrize("n", [2, 3])
def test_unet_replicas_equals_repeated_batch(tiny_sd, emu_backend, n):
    un = UNetModel(**TINY_UNET_CFG).eval()
    un.load_state_dict(sub_state_dict(tiny_sd, "model.diffusion_model."), strict=True)
    b, t, h, w = 2, 4, 8, 8
    inp = synth.synth_inputs(b, t, h, w, context_dim=TINY_UNET_CFG["context_dim"], seed=3)
    ctx = torch.cat([inp["cond"]] + [inp["uncond"] * (1.0 + 0.1 * k) for k in range(n - 1)], 0)
    ts = torch.tensor([601, 33])
    with torch.no_grad():
        full = un(None, ts.repeat(n), context=ctx, fs=inp["fs"].repeat(n),
                  x_parts=[inp["x_T"].repeat(n, 1, 1, 1, 1), inp["c_concat"].repeat(n, 1, 1, 1, 1)])
        un.reset_conditioning()
        shared = un(None, ts, context=ctx, fs=inp["fs"], x_parts=[inp["x_T"], inp["c_concat"]], replicas=n)
    assert shared.shape == full.shape == (n * b, 4, t, h, w)
    assert not torch.equal(full[:b], full[b:2 * b])                      # the passes do differ (through the context)
    assert torch.allclose(shared, full, rtol=0, atol=2e-5 * float(full.abs().max()))


def test_replicas_needs_matching_context(tiny_sd, emu_backend):
    un = UNetModel(**TINY_UNET_CFG).eval()
    un.load_state_dict(sub_state_dict(tiny_sd, "model.diffusion_model."), strict=True)
    inp = synth.synth_inputs(1, 4, 8, 8, context_dim=TINY_UNET_CFG["context_dim"], seed=4)
    with pytest.raises(ValueError):
        un(None, torch.tensor([5]), context=inp["cond"], fs=inp["fs"], x_parts=[inp["x_T"], inp["c_concat"]], replicas=2)


def test_apply_model_multi_shares_only_identical_concat(tiny_sd, emu_backend):
    """`uc` built from the SAME c_concat tensor (inference.py:213-214) takes the shared prefix; a different tensor with
    different values must not -- and both must equal the per-pass calls."""
    from test_two_clips import _pipeline
    model = _pipeline(tiny_sd, "cpu")
    model.use_hipgraph = False
    inp = synth.synth_inputs(1, 4, 8, 8, context_dim=96, seed=5)
    t = torch.tensor([401])
    seen = []
    real = model.model.diffusion_model.forward

    def spy(*a, **kw):
        seen.append(kw.get("replicas", 1))
        return real(*a, **kw)
    model.model.diffusion_model.forward = spy
    cond = {"c_crossattn": [inp["cond"]], "c_concat": [inp["c_concat"]]}
    with torch.no_grad():
        for other_concat, want in ((inp["c_concat"], 2), (inp["c_concat"] * 0.5, 1)):
            uc = {"c_crossattn": [inp["uncond"]], "c_concat": [other_concat]}
            model.reset_conditioning()
            seen.clear()
            e_c, e_u = model.apply_model_multi(inp["x_T"], t, [cond, uc], fs=inp["fs"])
            assert seen == [want]
            model.reset_conditioning()
            model.cfg_share = False
            r_c, r_u = model.apply_model_multi(inp["x_T"], t, [cond, uc], fs=inp["fs"])
            model.cfg_share = True
            scale = float(r_c.abs().max())
            assert torch.allclose(e_c, r_c, rtol=0, atol=2e-5 * scale) and torch.allclose(e_u, r_u, rtol=0, atol=2e-5 * scale)


def test_decode_core_groups_by_row_count_not_bytes():
    """decode_first_stage sends all B x T frames through one decoder call (the GEMM kernels address activations
    block-relatively); it falls back to groups only when a launch's ROW count would leave int32."""
    from tooncrafter_amd.lvdm.ddpm3d import LatentDiffusion

    class Dec:
        ch, calls = 128, []

        def decode_clip(self, z, refs, scale=1.0):
            self.calls.append((z.shape[0], None if refs is None else [r.shape[0] for r in refs]))
            return torch.zeros(z.shape[0], 3, z.shape[2], 1, 1)                       # (the real one returns 8h x 8w frames)

    class FS:
        decoder = Dec()
    m = LatentDiffusion.__new__(LatentDiffusion)
    torch.nn.Module.__init__(m)
    m.first_stage_model, m.scale_factor = FS(), 0.18215
    z = torch.zeros(3, 4, 16, 40, 64)                                    # 3 clips: 8 GB of level-0 activations, 7.9 M rows
    refs = [torch.zeros(3, 2, 8, 5, 5)]
    out = m.decode_core(z, ref_context=refs)
    assert out.shape == (3, 3, 16, 1, 1) and FS.decoder.calls == [(3, [3])]
    Dec.calls.clear()
    big = torch.zeros(1, 4, 16, 1280, 1024).expand(3, 4, 16, 1280, 1024)   # 1.3 G rows per clip: one clip per call
    m.decode_core(big, ref_context=refs)
    assert [c[0] for c in Dec.calls] == [1, 1, 1]


@pytest.mark.parametrize("n", [2, 3])
def test_unet_branches_equal_repeated_batch(tiny_sd, emu_backend, n):
    """`branches=True` (TC_CFG_STREAMS=1): the guided passes as n batch-b walks behind the shared prefix (on the GPU each
    on its own HIP stream) -- the list of their outputs is the plain batch-(n b) result, pass by pass; and the walk
    replays (does not recompute) what precedes the split."""
    un = UNetModel(**TINY_UNET_CFG).eval()
    un.load_state_dict(sub_state_dict(tiny_sd, "model.diffusion_model."), strict=True)
    b, t, h, w = 2, 4, 8, 8
    inp = synth.synth_inputs(b, t, h, w, context_dim=TINY_UNET_CFG["context_dim"], seed=3)
    ctx = torch.cat([inp["cond"]] + [inp["uncond"] * (1.0 + 0.1 * k) for k in range(n - 1)], 0)
    ts = torch.tensor([601, 33])
    be = ops.backend()
    calls = {"gn": 0}
    real_gn = be.groupnorm

    def counting_gn(*a, **kw):
        calls["gn"] += 1
        return real_gn(*a, **kw)
    with torch.no_grad():
        full = un(None, ts.repeat(n), context=ctx, fs=inp["fs"].repeat(n),
                  x_parts=[inp["x_T"].repeat(n, 1, 1, 1, 1), inp["c_concat"].repeat(n, 1, 1, 1, 1)])
        un.reset_conditioning()
        be.groupnorm = counting_gn
        try:
            one = un(None, ts, context=ctx[:b], fs=inp["fs"], x_parts=[inp["x_T"], inp["c_concat"]])
            gn_one, calls["gn"] = calls["gn"], 0
            un.reset_conditioning()
            outs = un(None, ts, context=ctx, fs=inp["fs"], x_parts=[inp["x_T"], inp["c_concat"]], replicas=n, branches=True)
            gn_br = calls["gn"]
        finally:
            del be.groupnorm
    assert isinstance(outs, list) and len(outs) == n and all(o.shape == (b, 4, t, h, w) for o in outs)
    assert torch.allclose(one, full[:b], rtol=0, atol=2e-5 * float(full.abs().max()))
    assert torch.allclose(torch.cat(outs, 0), full, rtol=0, atol=2e-5 * float(full.abs().max()))
    assert not torch.equal(outs[0], outs[1])
    # the shared prefix holds GroupNorms (input ResBlock, its temporal block, the first transformer's): passes 1.. skip them
    assert gn_one < gn_br < n * gn_one


def test_apply_model_multi_streams_mode_equals_per_pass_calls(tiny_sd, emu_backend):
    """TC_CFG_STREAMS=1 at the sampler boundary: same outputs as the unshared per-pass calls, and a second call with the
    same conditioning (the DDIM loop) reuses the static inputs."""
    from test_two_clips import _pipeline
    model = _pipeline(tiny_sd, "cpu")
    model.use_hipgraph = False
    inp = synth.synth_inputs(1, 4, 8, 8, context_dim=96, seed=6)
    cond = {"c_crossattn": [inp["cond"]], "c_concat": [inp["c_concat"]]}
    uc = {"c_crossattn": [inp["uncond"]], "c_concat": [inp["c_concat"]]}
    with torch.no_grad():
        model.cfg_streams = True
        for t in (torch.tensor([401]), torch.tensor([17])):
            e_c, e_u = model.apply_model_multi(inp["x_T"], t, [cond, uc], fs=inp["fs"])
            model.cfg_share, keep = False, model._cfg_state
            model._cfg_state = None
            r_c, r_u = model.apply_model_multi(inp["x_T"], t, [cond, uc], fs=inp["fs"])
            model.cfg_share, model._cfg_state = True, keep
            scale = float(r_c.abs().max())
            assert e_c.shape == r_c.shape and not torch.equal(e_c, e_u)
            assert torch.allclose(e_c, r_c, rtol=0, atol=2e-5 * scale) and torch.allclose(e_u, r_u, rtol=0, atol=2e-5 * scale)
