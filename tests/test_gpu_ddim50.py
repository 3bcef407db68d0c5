"""Parity on the configuration the metric is quoted on (BASELINE.json configs[1]): DDIM-50, CFG 7.5, eta 1,
guidance rescale 0.7, trailing spacing, full 16 x 40 x 64 latents, then the dual-reference decode to 320 x 512.

Loop under test: reference lvdm/models/samplers/ddim.py:135-279 (ddim_sampling / p_sample_ddim) driving
lvdm/modules/networks/openaimodel3d.py:548-603 through lvdm/models/ddpm3d.py:1243-1264, then
lvdm/models/ddpm3d.py:647-679 / autoencoder_dualref.py:489-527.

Three trajectories from the same x_T, the same conditioning and the same 50 INJECTED Gaussian draws (the reference
takes them from the device generator, which cannot be matched across implementations):
  * fp32 oracle (oracle/unet.py + oracle/sampler.py, PyTorch eager on this GPU)          -- the truth,
  * the same oracle under torch.autocast(bfloat16) (the reference runs under fp16 autocast, inference.py:323)
                                                                                         -- the noise floor,
  * the HIP path (batched-CFG B = 2 forwards, hipGraph replay, fused tc_ddim_step)       -- the thing under test.
CFG 7.5 amplifies the DIFFERENCE of two forwards and eta = 1 re-injects noise every step, so rounding differences grow
along the trajectory for ANY reduced-precision implementation; the bound is therefore relative to the floor measured in
the same run (<= 1.5 x floor), per step and for the final latent and the decoded pixels.

The two oracle trajectories cost ~200 s of PyTorch-eager time on the GPU (108 s fp32 + 91 s autocast), so they are
run ONCE by tests/golden/make_ddim50_golden.py ON THE MI355X (seeds only; `oracle_trajectories` below is the code it
runs) and committed as tests/golden/ddim50_oracle.npz: the fp32 trajectory at sampled positions, and the autocast
run's distance from it (the floor) per step, measured over the full tensors.  TC_LIVE_ORACLE=1 re-runs both oracles in
the test process and compares full tensors instead (the round-3 tables under profiles/ were made that way).

The per-step table goes to gpurun_out/ddim50_parity_<mode>.txt (copied to profiles/ by hand).
"""
import os

import numpy as np
import pytest
import torch

import fullsize_cases as fc
from conftest import ROOT, rel_l2, sub_state_dict
from tooncrafter_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda"
S = 50
# HIP error / floor, per step.  Both numerators are single draws of a random rounding process, so individual steps
# scatter around the mean ratio; the assertion is on the median and on a looser per-step cap.
MEDIAN_RATIO, STEP_RATIO, FINAL_RATIO, PIXEL_RATIO = 1.25, 1.75, 1.5, 1.5


def _noises():
    return [torch.randn(1, 4, fc.T, fc.H, fc.W, generator=torch.Generator().manual_seed(300 + i)).to(DEV)
            for i in range(S)]


GOLDEN = os.path.join(ROOT, "tests", "golden", "ddim50_oracle.npz")
N_X0, N_FINAL, N_PIX = 8192, 16384, 131072          # sampled positions per step / of the final latent / of the pixels


def oracle_trajectories(full_model, inp):
    """fp32 and bf16-autocast oracle trajectories + decodes on the GPU (full tensors)."""
    from oracle import decoder as odec
    from oracle import sampler as osamp
    from oracle import unet as ounet
    torch.backends.cuda.matmul.allow_tf32 = False
    sd_all = {n: p.detach() for n, p in full_model.named_parameters()}
    usd = sub_state_dict(sd_all, "model.diffusion_model.")
    dsd = sub_state_dict(sd_all, "first_stage_model.decoder.")
    dev = lambda k: inp[k].to(DEV)
    noises = _noises()
    sched = osamp.make_schedule_buffers()          # CPU tables: the loop reads scalars from them (as scripts/noise_floor.py)
    cc = dev("c_concat")
    refs = [r.to(DEV) for r in inp["refs"]]

    def run(autocast):
        def unet(x, t, c):
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                return ounet.unet_forward(usd, fc.UNET_CFG, torch.cat([x, cc], 1), t, c, dev("fs")).float()
        x0s = []
        with torch.no_grad():
            fin = osamp.ddim_sample(unet, dev("x_T"), dev("cond"), dev("uncond"), S, fc.ETA, fc.CFG, fc.RESCALE, sched,
                                    noise_fn=lambda i: noises[i],
                                    step_callback=lambda i, img, p: x0s.append(p.clone()))
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                pix = odec.decode_first_stage(dsd, fin, refs).float()
        return dict(final=fin, x0s=x0s, pix=pix)
    return dict(fp32=run(False), bf16=run(True), noises=noises)


def sample_positions(x0_numel, final_numel, pix_numel):
    return (fc.sample_idx(x0_numel, N_X0, 11), fc.sample_idx(final_numel, N_FINAL, 12), fc.sample_idx(pix_numel, N_PIX, 13))


class Reference:
    """What the HIP path is compared with: `err_*(tensor)` = rel-L2 against the fp32 oracle, `floor_*` = the autocast
    oracle's distance from it."""

    def __init__(self, live=None):
        self.live = live
        if live is None:
            g = np.load(GOLDEN)
            self.x0 = torch.from_numpy(g["x0_fp32"])
            self.final, self.pix = torch.from_numpy(g["final_fp32"]), torch.from_numpy(g["pix_fp32"])
            self.floor_x0, self.floor_final, self.floor_pix = g["floor_x0"].tolist(), float(g["floor_final"]), float(g["floor_pix"])
            self.idx = None
        else:
            ref, flo = live["fp32"], live["bf16"]
            self.floor_x0 = [rel_l2(flo["x0s"][i], ref["x0s"][i]) for i in range(S)]
            self.floor_final, self.floor_pix = rel_l2(flo["final"], ref["final"]), rel_l2(flo["pix"], ref["pix"])

    def _idx(self, hip):
        if self.idx is None:
            self.idx = sample_positions(hip["x0s"][0][:1].numel(), hip["final"][:1].numel(), hip["pix"][:1].numel())
        return self.idx

    def err_x0(self, hip, i):
        if self.live is not None:
            return rel_l2(hip["x0s"][i][:1], self.live["fp32"]["x0s"][i])
        return rel_l2(hip["x0s"][i][:1].reshape(-1)[self._idx(hip)[0].to(DEV)].cpu(), self.x0[i])

    def err_final(self, hip):
        if self.live is not None:
            return rel_l2(hip["final"][:1], self.live["fp32"]["final"])
        return rel_l2(hip["final"][:1].reshape(-1)[self._idx(hip)[1].to(DEV)].cpu(), self.final)

    def err_pix(self, hip):
        if self.live is not None:
            return rel_l2(hip["pix"][:1], self.live["fp32"]["pix"])
        return rel_l2(hip["pix"][:1].reshape(-1)[self._idx(hip)[2].to(DEV)].cpu(), self.pix)


@pytest.fixture(scope="module")
def oracle_runs(full_model, inp):
    if os.environ.get("TC_LIVE_ORACLE") == "1":
        live = oracle_trajectories(full_model, inp)
        return dict(ref=Reference(live), noises=live["noises"], source="both oracles run in this process")
    return dict(ref=Reference(), noises=_noises(), source="tests/golden/ddim50_oracle.npz (oracles run on an MI355X by make_ddim50_golden.py)")


def _hip_run(full_model, inp, noises, batch=1):
    from tooncrafter_amd.lvdm import ddim as my_ddim
    rep = lambda t: torch.cat([t.to(DEV)] * batch, 0)
    cond = {"c_crossattn": [rep(inp["cond"])], "c_concat": [rep(inp["c_concat"])]}
    uc = {"c_crossattn": [rep(inp["uncond"])], "c_concat": [rep(inp["c_concat"])]}
    it = iter(noises)
    old = my_ddim.noise_like
    my_ddim.noise_like = lambda shape, device, repeat=False: torch.cat([next(it)] * batch, 0)
    x0s = []
    full_model._cfg_state = None
    try:
        with torch.no_grad():
            out, _ = my_ddim.DDIMSampler(full_model).sample(
                S=S, conditioning=cond, batch_size=batch, shape=(4, fc.T, fc.H, fc.W), verbose=False,
                unconditional_guidance_scale=fc.CFG, unconditional_conditioning=uc, eta=fc.ETA, fs=rep(inp["fs"]),
                timestep_spacing="uniform_trailing", guidance_rescale=fc.RESCALE, x_T=rep(inp["x_T"]),
                img_callback=lambda p, i: x0s.append(p.clone()))
            refs = [rep(r) for r in inp["refs"]]
            pix = full_model.decode_first_stage(out, ref_context=refs)
    finally:
        my_ddim.noise_like = old
        full_model._cfg_state = None
    return dict(final=out, x0s=x0s, pix=pix.float())


def _report(mode, hip, orc, extra=""):
    ref = orc["ref"]
    rows, ratios = [], []
    for i in range(S):
        f, h = ref.floor_x0[i], ref.err_x0(hip, i)
        ratios.append(h / f)
        rows.append(f"{i:4d}  {f:.3e}  {h:.3e}  {h / f:5.2f}")
    ff, hf = ref.floor_final, ref.err_final(hip)
    fp, hp = ref.floor_pix, ref.err_pix(hip)
    med = sorted(ratios)[S // 2]
    text = "\n".join([
        f"# DDIM-{S} CFG {fc.CFG} eta {fc.ETA} rescale {fc.RESCALE}, 16x40x64 latents, mode = {mode}{extra}",
        f"# {torch.cuda.get_device_name(0)}; rel-L2 of pred_x0 against the fp32 oracle trajectory (same injected noise); oracle: {orc['source']}",
        "step  floor(bf16-autocast oracle)  HIP path  ratio", *rows,
        f"final latent: floor {ff:.3e}  HIP {hf:.3e}  ratio {hf / ff:.2f}",
        f"decoded pixels (16 x 320 x 512): floor {fp:.3e}  HIP {hp:.3e}  ratio {hp / fp:.2f}",
        f"median per-step ratio {med:.2f}, max {max(ratios):.2f}"])
    print(text)
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, f"ddim50_parity_{mode}.txt"), "w") as f:
        f.write(text + "\n")
    assert torch.isfinite(hip["final"]).all() and torch.isfinite(hip["pix"]).all()
    assert med <= MEDIAN_RATIO, f"median per-step ratio {med:.2f}"
    assert max(ratios) <= STEP_RATIO, f"worst step ratio {max(ratios):.2f}"
    assert hf <= FINAL_RATIO * ff and hp <= PIXEL_RATIO * fp


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("mode", ["bf16", "fp8_linear"])
def test_ddim50_full_size_vs_oracle(full_model, inp, oracle_runs, mode):
    """Default bf16 path and the MXFP8 qkv / GEGLU routing (BASELINE.json configs[4], TC_FP8=1) under the SAME bound."""
    be = ops.backend()
    old = be.fp8
    be.fp8 = "linear" if mode == "fp8_linear" else None
    try:
        hip = _hip_run(full_model, inp, oracle_runs["noises"])
    finally:
        be.fp8 = old
    _report(mode, hip, oracle_runs)


@pytest.mark.timeout(1800)
def test_ddim50_batched_decode_geometry(full_model, inp, oracle_runs):
    """BASELINE.json configs[3] geometry: two clips per sampler call (batched-CFG B = 4 forwards) and ONE decode call
    over both clips.  Both clips carry the reference inputs, so each must satisfy the single-clip bound and the two
    must agree with each other bit for bit (no cross-sample term anywhere in the path)."""
    hip = _hip_run(full_model, inp, oracle_runs["noises"], batch=2)
    assert torch.equal(hip["final"][0], hip["final"][1]) and torch.equal(hip["pix"][0], hip["pix"][1])
    _report("batch2", hip, oracle_runs, extra=" (2 clips per call, one decode call)")
