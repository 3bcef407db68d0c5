"""Full-size parity of the HIP path against the fp32 CPU oracle (BASELINE.json configs[1] shapes:
320-channel UNet, 16 frames, 40x64 latents; VideoDecoder to 320x512 with the 10240 x 20480 reference
attention, 256-row tile convolutions, split-K).

The oracle runs for minutes per case on a CPU, so its outputs are committed once
(tests/golden/fullsize_oracle.npz, written by tests/golden/make_fullsize_golden.py from seeds only)
and replayed here; TC_LIVE_ORACLE=1 additionally re-runs the oracle on this host for the UNet case.

Stated tolerances (bf16 weights/activations, fp32 accumulation, statistics, softmax and DDIM state), calibrated
in profiles/r02_noise_floor.txt against the noise floor of the SAME oracle code run under torch bf16 autocast
on the GPU -- how the reference itself runs (fp16 autocast, inference.py:323) -- per SURVEY.md 8d
(bound <= 1.5 x floor):
                               floor (autocast oracle)   HIP path measured   bound here
  one UNet forward             1.72e-2                   1.44e-2             2.0e-2 (1.16 x floor), cosine >= 0.9995
  decoder output               1.35e-2                   1.30e-2             2.0e-2 (1.48 x floor)
  decoder stages               4.8e-3 ... 1.46e-2        4.7e-3 ... 1.44e-2  1.5 x the stage's floor
  DDIM-3 CFG 7.5 pred_x0       7.1e-2 / 6.3e-2 / 4.5e-2  6.1 / 5.4 / 3.9e-2  1.25 x the step's floor
  DDIM-3 final latent          4.5e-2                    3.9e-2              5.7e-2 (1.25 x floor)
(CFG 7.5 multiplies the DIFFERENCE of two forwards, so the trajectory floor is ~4x the single-forward one.)
"""
import os

import numpy as np
import pytest
import torch

import fullsize_cases as fc
from conftest import ROOT, rel_l2
from tooncrafter_amd import ops, synth

pytestmark = pytest.mark.gpu
DEV = "cuda"

UNET_REL, UNET_COS = 2.0e-2, 0.9995
DEC_REL = 2.0e-2
DEC_STAGE_FLOOR = {"mid": 4.78e-3, "level3": 4.91e-3, "level2": 7.98e-3, "level1": 1.086e-2, "level0": 1.463e-2}
DDIM_X0_FLOOR, DDIM_FINAL_FLOOR = (7.144e-2, 6.283e-2, 4.530e-2), 4.523e-2


def cosine(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm()))


@pytest.mark.timeout(1500)
def test_unet_full_size_vs_oracle(full_model, golden, inp):
    un = full_model.model.diffusion_model
    ts = torch.tensor([fc.UNET_T], device=DEV)
    with torch.no_grad():
        y = un(None, ts, context=inp["cond"].to(DEV), fs=inp["fs"].to(DEV),
               x_parts=[inp["x_T"].to(DEV), inp["c_concat"].to(DEV)]).cpu()
    ref = torch.from_numpy(golden["unet_y"])
    e, c = rel_l2(y, ref), cosine(y, ref)
    print(f"full-size UNet (B=1, t={fc.UNET_T}) HIP vs fp32 CPU oracle: rel-L2 {e:.3e}, cosine {c:.6f}")
    assert torch.isfinite(y).all() and y.shape == ref.shape
    assert e <= UNET_REL and c >= UNET_COS
    if os.environ.get("TC_LIVE_ORACLE") == "1":
        from conftest import sub_state_dict
        from oracle import unet as ounet
        usd = sub_state_dict(fc.full_state_dict(("model.diffusion_model.",)), "model.diffusion_model.")
        with torch.no_grad():
            live = ounet.unet_forward(usd, fc.UNET_CFG, torch.cat([inp["x_T"], inp["c_concat"]], 1),
                                      torch.tensor([fc.UNET_T]), inp["cond"], inp["fs"])
        print(f"  live oracle on this host vs committed golden: rel-L2 {rel_l2(live, ref):.3e}")
        assert rel_l2(live, ref) < 1e-4


@pytest.mark.timeout(1500)
def test_unet_full_size_shared_cfg_prefix(full_model, golden, inp):
    """Batched guidance with the shared prefix (lvdm/common.py: CfgShare; what apply_model_multi issues): `replicas=2` on
    single-copy inputs against the plain B = 2 call on repeated inputs, and against the fp32 oracle.  The prefix runs at
    half the rows, where the launch heuristics pick other kernels (the weight-stationary projections need M >= 64K), so
    the two are bf16 realisations of the same arithmetic, each ~1.4e-2 from the oracle."""
    un = full_model.model.diffusion_model
    ctx2 = torch.cat([inp["cond"], inp["uncond"]]).to(DEV)
    x, cc, fs = inp["x_T"].to(DEV), inp["c_concat"].to(DEV), inp["fs"].to(DEV)
    ts = torch.tensor([fc.UNET_T], device=DEV)
    with torch.no_grad():
        full = un(None, ts.repeat(2), context=ctx2, fs=fs.repeat(2), x_parts=[x.repeat(2, 1, 1, 1, 1), cc.repeat(2, 1, 1, 1, 1)]).clone()
        shared = un(None, ts, context=ctx2, fs=fs, x_parts=[x, cc], replicas=2)
    ref = torch.from_numpy(golden["unet_y"])
    e_full, e_shared = rel_l2(full[:1].cpu(), ref), rel_l2(shared[:1].cpu(), ref)
    d = [rel_l2(shared[i], full[i]) for i in range(2)]
    print(f"full-size UNet, batched guidance: repeated inputs vs oracle {e_full:.3e}, shared prefix vs oracle {e_shared:.3e}; "
          f"shared vs repeated {d[0]:.3e} / {d[1]:.3e}; cond vs uncond pass {rel_l2(full[0], full[1]):.3e}")
    assert torch.isfinite(shared).all() and shared.shape == full.shape
    assert e_shared <= UNET_REL and e_full <= UNET_REL
    assert max(d) <= 2e-2


@pytest.mark.timeout(1500)
def test_unet_full_size_guided_passes_on_streams(full_model, golden, inp):
    """TC_CFG_STREAMS=1 (UNetModel._forward_branches): behind the shared prefix the two guided passes are two batch-1
    walks on their own HIP streams.  Against the batch-2 walk (another bf16 realisation: other tile families at half the
    rows) and the fp32 oracle; then through apply_model_multi, where the fork / join is captured into the hipGraph --
    eager call, capturing call and two replays must agree bit for bit (a pass reading a tensor of the other stream too
    early, or K/V projected behind the fork, would show here)."""
    un = full_model.model.diffusion_model
    ctx2 = torch.cat([inp["cond"], inp["uncond"]]).to(DEV)
    x, cc, fs = inp["x_T"].to(DEV), inp["c_concat"].to(DEV), inp["fs"].to(DEV)
    ts = torch.tensor([fc.UNET_T], device=DEV)
    with torch.no_grad():
        shared = un(None, ts, context=ctx2, fs=fs, x_parts=[x, cc], replicas=2).clone()
        un.reset_conditioning()
        outs = un(None, ts, context=ctx2, fs=fs, x_parts=[x, cc], replicas=2, branches=True)
        outs = [o.clone() for o in outs]
        again = un(None, ts, context=ctx2, fs=fs, x_parts=[x, cc], replicas=2, branches=True)
    torch.cuda.synchronize()
    ref = torch.from_numpy(golden["unet_y"])
    e = rel_l2(outs[0].cpu(), ref)
    d = [rel_l2(outs[i][0], shared[i]) for i in range(2)]
    print(f"full-size UNet, guided passes on streams: cond pass vs oracle {e:.3e}; vs the batch-2 walk {d[0]:.3e} / {d[1]:.3e}")
    assert len(outs) == 2 and all(o.shape == (1, *shared.shape[1:]) and torch.isfinite(o).all() for o in outs)
    assert e <= UNET_REL and max(d) <= 2e-2
    assert all(torch.equal(a, b) for a, b in zip(outs, again))             # deterministic across calls (no race)

    cond = {"c_crossattn": [inp["cond"].to(DEV)], "c_concat": [cc]}
    uc = {"c_crossattn": [inp["uncond"].to(DEV)], "c_concat": [cc]}
    old = (full_model.cfg_streams, full_model.use_hipgraph, full_model._cfg_state)
    full_model.cfg_streams, full_model.use_hipgraph, full_model._cfg_state = True, True, None
    try:
        with torch.no_grad():
            runs = [[o.clone() for o in full_model.apply_model_multi(x, ts, [cond, uc], fs=fs)] for _ in range(4)]
        torch.cuda.synchronize()
        assert full_model._cfg_state["graph"] is not None
        for r in runs[1:]:
            assert torch.equal(r[0], runs[0][0]) and torch.equal(r[1], runs[0][1])
        assert torch.equal(runs[0][0], outs[0]) and torch.equal(runs[0][1], outs[1])
    finally:
        full_model.cfg_streams, full_model.use_hipgraph, full_model._cfg_state = old
        full_model.reset_conditioning()


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("fp8", [None, "linear"])
def test_ddim3_full_size_vs_oracle(full_model, golden, inp, fp8):
    """3-step CFG-7.5 DDIM (rescale 0.7, eta 1, trailing) with injected noise: batched-CFG B=2 UNet calls,
    hipGraph replay from the second step on, fused tc_ddim_step.  fp8 = "linear": the same trajectory with the
    qkv / GEGLU projections on the MXFP8 kernel (BASELINE.json configs[4], TC_FP8=1) under the SAME bound
    (measured 7.5e-2 / 6.6e-2 / 4.9e-2, final 4.8e-2 = 1.05-1.07 x the bf16-autocast floor)."""
    from tooncrafter_amd.lvdm import ddim as my_ddim
    be = ops.backend()
    old_fp8 = be.fp8
    be.fp8 = fp8
    full_model._cfg_state = None                  # a captured graph replays the kernels it recorded
    noises = iter([n.to(DEV) for n in inp["noises"]])
    dev = lambda k: inp[k].to(DEV)
    cond = {"c_crossattn": [dev("cond")], "c_concat": [dev("c_concat")]}
    uc = {"c_crossattn": [dev("uncond")], "c_concat": [dev("c_concat")]}
    old = my_ddim.noise_like
    my_ddim.noise_like = lambda shape, device, repeat=False: next(noises)
    x0s = []
    try:
        with torch.no_grad():
            out, _ = my_ddim.DDIMSampler(full_model).sample(
                S=fc.DDIM_STEPS, conditioning=cond, batch_size=1, shape=(4, fc.T, fc.H, fc.W), verbose=False,
                unconditional_guidance_scale=fc.CFG, unconditional_conditioning=uc, eta=fc.ETA, fs=dev("fs"),
                timestep_spacing="uniform_trailing", guidance_rescale=fc.RESCALE, x_T=dev("x_T"),
                img_callback=lambda p, i: x0s.append(p.clone()))
    finally:
        my_ddim.noise_like = old
        be.fp8 = old_fp8
        full_model._cfg_state = None
    errs = [rel_l2(p.cpu(), torch.from_numpy(golden[f"ddim_pred_x0_{i}"])) for i, p in enumerate(x0s)]
    final = rel_l2(out.cpu(), torch.from_numpy(golden["ddim_final"]))
    print(f"full-size DDIM-3 CFG 7.5 (fp8 = {fp8}) vs fp32 CPU oracle: pred_x0 rel-L2 per step", [f"{e:.3e}" for e in errs],
          f"final latent {final:.3e}")
    assert torch.isfinite(out).all()
    assert all(e <= 1.25 * f for e, f in zip(errs, DDIM_X0_FLOOR)) and final <= 1.25 * DDIM_FINAL_FLOOR


@pytest.mark.timeout(1500)
def test_decoder_full_size_two_clips_one_call(full_model, golden, inp):
    """BASELINE.json configs[3] (perframe_ae=False) at 320x512: two DIFFERENT clips through ONE decode_first_stage call
    -- B * T = 32 frames per launch, level-0 activations of 2.7 GB, beyond what 31-bit tensor-relative buffer offsets
    could address (the GEMM kernels address block-relatively, csrc/gemm_common.h).  The reference raises on this call
    (lvdm/models/ddpm3d.py:656-657).  Three statements:
      * parity: clip 0 of the batched call against the fp32 oracle golden, under the single-clip bound;
      * no cross-clip term: the same clip in both batch slots of one call gives bit-identical halves;
      * against one-clip-per-call decodes: a 32-frame launch picks other tile families / split-K / GroupNorm chunkings
        than a 16-frame one (the heuristics look at the row count), so this compares two bf16 REALISATIONS of the same
        arithmetic -- each ~1.3e-2 from the fp32 oracle -- and must stay inside that scale (the addressing itself is pinned
        bit for bit by test_gpu_ops.py::test_gemm_activation_beyond_2gib; the tiny decoder, whose launches keep their
        geometry when batched, is bit-identical: test_gpu_models.py)."""
    z1 = inp["z_dec"].to(DEV)
    z2 = (torch.roll(z1, 3, dims=4) * 0.9).contiguous()
    r1 = [r.to(DEV) for r in inp["refs"]]
    r2 = [(torch.flip(r, dims=(-1,)) * 1.1).contiguous() for r in r1]
    with torch.no_grad():
        y1 = full_model.decode_first_stage(z1, ref_context=r1).clone()
        y2 = full_model.decode_first_stage(z2, ref_context=r2).clone()
        yy = full_model.decode_first_stage(torch.cat([z1, z2], 0), ref_context=[torch.cat([a, b], 0) for a, b in zip(r1, r2)])
        tw = full_model.decode_first_stage(torch.cat([z1, z1], 0), ref_context=[torch.cat([a, a], 0) for a in r1])
    assert tuple(yy.shape) == (2, 3, z1.shape[2], 320, 512) and torch.isfinite(yy).all()
    assert not torch.equal(y1, y2)
    flat = yy[0].reshape(-1)
    got = flat[fc.sample_idx(flat.numel(), fc.N_OUT, 1).to(DEV)].cpu()
    e_or = rel_l2(got, torch.from_numpy(golden["dec16_out"]))
    e1, e2 = rel_l2(yy[0], y1[0]), rel_l2(yy[1], y2[0])
    worst = float((yy[0] - y1[0]).abs().max())
    print(f"two clips in one decode call: clip 0 vs fp32 oracle {e_or:.3e}; vs one clip per call rel-L2 {e1:.3e} / {e2:.3e} "
          f"(max-abs {worst:.3e}); the clips themselves differ by {rel_l2(y1, y2):.3e}")
    assert e_or <= DEC_REL
    assert torch.equal(tw[0], tw[1])
    assert e1 <= 2e-2 and e2 <= 2e-2 and worst < 0.5


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("tag", ["dec16", "dec14"])
def test_decoder_full_size_vs_oracle(full_model, golden, inp, tag):
    """VideoDecoder at 40x64 latents -> 320x512: 16 frames, and the 14-frame re-decode that reuses the cached
    reference K/V (inference.py:264-270).  Output and every stage (mid block, each level after its
    reference fusion / combiner) at the committed sample positions."""
    z = inp["z_dec"] if tag == "dec16" else inp["z_dec"][:, :, fc.IDX14].contiguous()
    refs = [r.to(DEV) for r in inp["refs"]]
    stages = {}

    def probe(name, act):
        flat = fc.nchw_flat_from_rows(act.rows, act.frames, act.h, act.w)
        idx = fc.sample_idx(flat.numel(), fc.N_PROBE, 2).to(DEV)
        stages[name] = flat[idx].float().cpu()
    dec = full_model.first_stage_model.decoder
    with torch.no_grad():
        if tag == "dec14":                      # as in the pipeline: the 16-frame decode ran first with the same refs
            dec.decode_clip(inp["z_dec"].to(DEV), refs, scale=1.0 / 0.18215)
        y = dec.decode_clip(z.to(DEV), refs, scale=1.0 / 0.18215, probe=probe)
    flat = y.reshape(-1)
    got = flat[fc.sample_idx(flat.numel(), fc.N_OUT, 1).to(DEV)].cpu()
    e_out = rel_l2(got, torch.from_numpy(golden[f"{tag}_out"]))
    e_st = {n: rel_l2(stages[n], torch.from_numpy(golden[f"{tag}_{n}"])) for n in fc.PROBES}
    norm_ratio = float(y.double().norm()) / float(golden[f"{tag}_out_norm"])
    print(f"full-size decoder {tag} vs fp32 CPU oracle: out rel-L2 {e_out:.3e} (|y|/|ref| {norm_ratio:.4f}); stages",
          {n: f"{e:.3e}" for n, e in e_st.items()})
    assert torch.isfinite(y).all() and tuple(y.shape) == (1, 3, z.shape[2], 320, 512)
    assert e_out <= DEC_REL and abs(norm_ratio - 1.0) < 5e-3
    assert all(e_st[n] <= 1.5 * DEC_STAGE_FLOOR[n] for n in fc.PROBES), e_st


# First-stage encoder at the full size (row f1; the reference call is scripts/evaluation/inference.py:164-178 ->
# lvdm/models/autoencoder.py:100-110 -> lvdm/modules/networks/ae_modules.py:432-475).  The golden holds the REAL
# reference's fp32 values at sampled positions (tests/golden/make_encoder_fullsize_golden.py; the oracle reproduces
# them exactly, profiles/r04_encoder_fullsize_golden_vs_reference.txt).  Bounds: the bf16 path's distance measured on the
# MI355X (profiles/r04_encoder_fullsize_parity.txt) x 1.5.
ENC_BOUND = dict(mean=1.95e-2, logvar=2.3e-2, hid0=8.5e-3, hid1=1.3e-2, hid2=1.7e-2, hid3=2.0e-2, hid4=4.2e-3)
# measured: mean 1.30e-2, logvar 1.51e-2, hid0 5.66e-3, hid1 8.67e-3, hid2 1.14e-2, hid3 1.31e-2, hid4 2.78e-3


@pytest.mark.timeout(900)
def test_encoder_full_size_vs_reference_golden(full_model):
    g = np.load(fc.ENC_GOLDEN_FILE)
    x = fc.encoder_frames().to(DEV)
    with torch.no_grad():
        post, hidden = full_model.first_stage_model.encode(x, return_hidden_states=True)
    got = dict(mean=post.mean, logvar=post.logvar, **{f"hid{i}": h for i, h in enumerate(hidden)})
    errs = {}
    for name, t in got.items():
        assert tuple(t.shape) == tuple(g[name + "_shape"]), (name, t.shape)
        flat = t.float().reshape(-1)
        idx = fc.sample_idx(flat.numel(), fc.N_ENC, fc.ENC_SEEDS[name]).to(DEV)
        errs[name] = rel_l2(flat[idx].cpu(), torch.from_numpy(g[name]))
        nr = float(t.double().norm()) / float(g[name + "_norm"])
        assert abs(nr - 1.0) < 1e-2, (name, nr)
    text = "first-stage encoder 16 x 3 x 320 x 512, HIP path vs the real reference (fp32 CPU) at 65536 sampled positions: " + \
        ", ".join(f"{k} {v:.3e}" for k, v in errs.items())
    print(text)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "encoder_fullsize_parity.txt"), "w") as f:
        f.write(text + "\n")
    assert all(torch.isfinite(t).all() for t in got.values())
    assert all(errs[k] <= ENC_BOUND[k] for k in errs), errs


# MXFP8 GEMM path (BASELINE.json configs[4]): its OWN parity bounds.  The kernels are exact against the MX
# restatement (tests/test_gpu_fp8.py); what is bounded here is the error the 8-bit operand format introduces into
# one full-size UNet forward, against the same fp32 CPU oracle golden (profiles/r02_fp8_error_by_layer_class.txt):
#   TC_FP8=1 ("linear": the qkv / GEGLU projections, N >= 2 K: 51 launches)  1.91e-2 -> bound 2.4e-2 (1.4 x the
#                                                          bf16-autocast floor of 1.72e-2: inside SURVEY 8d's 1.5 x)
#   TC_FP8=all (+ 3x3 / temporal convolutions, 143 launches)                 1.40e-1 -> bound 1.8e-1
UNET_FP8 = {"linear": (2.4e-2, 0.9995, 45), "all": (1.8e-1, 0.985, 130)}


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("mode", ["linear", "all"])
def test_unet_full_size_fp8(full_model, golden, inp, mode):
    be = ops.backend()
    un = full_model.model.diffusion_model
    ts = torch.tensor([fc.UNET_T], device=DEV)
    args = dict(context=inp["cond"].to(DEV), fs=inp["fs"].to(DEV))
    parts = [inp["x_T"].to(DEV), inp["c_concat"].to(DEV)]
    bound, cos_min, min_calls = UNET_FP8[mode]
    with torch.no_grad():
        y16 = un(None, ts, x_parts=parts, **args).cpu()
        old, c0 = be.fp8, dict(be.fp8_calls)
        be.fp8 = mode
        try:
            y8 = un(None, ts, x_parts=parts, **args).cpu()
        finally:
            be.fp8 = old
    calls = {k: be.fp8_calls[k] - c0[k] for k in c0}
    ref = torch.from_numpy(golden["unet_y"])
    e8, c8, e16 = rel_l2(y8, ref), cosine(y8, ref), rel_l2(y16, ref)
    print(f"full-size UNet, TC_FP8={mode}: MXFP8 on {calls['mx']} of {calls['mx'] + calls['bf16']} GEMM launches: rel-L2 vs fp32 "
          f"oracle {e8:.3e} (bf16 path {e16:.3e}), cosine {c8:.6f}; fp8 vs bf16 path {rel_l2(y8, y16):.3e}")
    assert calls["mx"] > min_calls and torch.isfinite(y8).all()
    assert e8 <= bound and c8 >= cos_min


DEC_FP8_REL = 1.6e-1       # measured 1.30e-1 (bf16 path: 1.30e-2): why TC_FP8 leaves the decoder in bf16 by default


@pytest.mark.timeout(1500)
def test_decoder_full_size_fp8(full_model, golden, inp):
    """The 16-frame decode with the eligible convolutions on the MXFP8 kernel (eager: a captured graph would replay
    whatever kernels it recorded), against the same fp32 oracle samples as the bf16 test."""
    be = ops.backend()
    dec = full_model.first_stage_model.decoder
    refs = [r.to(DEV) for r in inp["refs"]]
    old, old_graph, old_dec, c0 = be.fp8, dec.use_hipgraph, be.fp8_decoder, dict(be.fp8_calls)
    be.fp8, dec.use_hipgraph = "all", False
    try:
        with torch.no_grad():
            dec.decode_clip(inp["z_dec"].to(DEV), refs, scale=1.0 / 0.18215)
            assert be.fp8_calls["mx"] == c0["mx"], "the decoder must stay on the bf16 kernels unless TC_FP8_DECODER=1"
            c0 = dict(be.fp8_calls)
            be.fp8_decoder = True
            y = dec.decode_clip(inp["z_dec"].to(DEV), refs, scale=1.0 / 0.18215)
    finally:
        be.fp8, dec.use_hipgraph, be.fp8_decoder = old, old_graph, old_dec
    calls = {k: be.fp8_calls[k] - c0[k] for k in c0}
    flat = y.reshape(-1)
    got = flat[fc.sample_idx(flat.numel(), fc.N_OUT, 1).to(DEV)].cpu()
    e = rel_l2(got, torch.from_numpy(golden["dec16_out"]))
    print(f"full-size decoder, MXFP8 on {calls['mx']} of {calls['mx'] + calls['bf16']} GEMM launches: out rel-L2 vs fp32 oracle {e:.3e}")
    assert calls["mx"] > 20 and torch.isfinite(y).all()
    assert e <= DEC_FP8_REL
