"""Module- and pipeline-level parity of the HIP path on the MI355X.

Three references, in increasing independence:
  (1) the same module mirror run on the PyTorch statement of the operator contract
      (tests/emu_ops.py, on the GPU) -- identical rounding points, so only fp32 summation
      order differs: tight bound;
  (2) the CPU oracle (oracle/, fp32);
  (3) the committed goldens produced by the real reference (tests/golden/*.npz).
Stated tolerances (bf16 activations and weights, fp32 accumulation/statistics):
  one UNet / decoder forward vs fp32 reference: rel-L2 <= 3e-2;
  DDIM trajectory with CFG 7.5: rel-L2 <= 0.15 (CFG amplifies the per-forward noise ~10x);
  HIP vs emulated contract at module level: rel-L2 <= 3e-2 (a 1-ulp bf16 rounding flip early in a
  100-operator chain propagates like fresh rounding noise, so this is the same floor as vs fp32;
  the tight per-operator bounds live in test_gpu_ops.py).
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import (FULL_UNET_CFG, GOLDEN, TINY_DD_CFG, TINY_UNET_CFG, load_golden, rel_l2, sub_state_dict)
from emu_ops import EmuOps
from tooncrafter_amd import ops, synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _with_backend(backend, fn):
    prev = ops.set_backend(backend)
    try:
        return fn()
    finally:
        ops.set_backend(prev)


@pytest.fixture(scope="module")
def hip():
    from tooncrafter_amd.ops import HipOps
    return HipOps()


@pytest.fixture(scope="module")
def tiny_unet(tiny_sd):
    from tooncrafter_amd.lvdm.openaimodel3d import UNetModel
    un = UNetModel(**TINY_UNET_CFG).eval()
    un.load_state_dict(sub_state_dict(tiny_sd, "model.diffusion_model."), strict=True)
    return un.to(DEV)


@pytest.fixture(scope="module")
def tiny_decoder(tiny_sd):
    from tooncrafter_amd.lvdm.autoencoder_dualref import VideoDecoder
    vd = VideoDecoder(**TINY_DD_CFG).eval()
    vd.load_state_dict(sub_state_dict(tiny_sd, "first_stage_model.decoder."), strict=True)
    return vd.to(DEV)


def test_unet_tiny_vs_reference_golden(hip, tiny_unet):
    g = load_golden("unet_tiny.npz")
    args = (torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["timesteps"]).to(DEV))
    kw = dict(context=torch.from_numpy(g["context"]).to(DEV), fs=torch.from_numpy(g["fs"]).to(DEV))
    with torch.no_grad():
        y = _with_backend(hip, lambda: tiny_unet(*args, **kw))
        y_emu = _with_backend(EmuOps(), lambda: tiny_unet(*args, **kw))
    ref = torch.from_numpy(g["y"])
    e_ref, e_emu = rel_l2(y.cpu(), ref), rel_l2(y.cpu(), y_emu.cpu())
    print(f"tiny UNet: vs reference golden {e_ref:.3e}; vs emulated contract {e_emu:.3e}; "
          f"emulated contract vs golden {rel_l2(y_emu.cpu(), ref):.3e}")
    assert torch.isfinite(y).all()
    assert e_ref < 3.5e-2 and e_emu < 3.5e-2        # 1.5 x the measured 2.31e-2 / 2.28e-2 (profiles/r04_tiny_model_parity.txt)


def test_unet_deterministic_and_batch_consistent(hip, tiny_unet):
    """Properties: (a) bit-identical across runs (no atomics); (b) a sample's output does not depend
    on what else is in the batch -- the basis of running cond+uncond as one B=2 call."""
    inp = synth.synth_inputs(2, 4, 8, 8, context_dim=96, seed=3)
    x = torch.cat([inp["x_T"], inp["c_concat"]], 1).to(DEV)
    ts = torch.tensor([601, 601], device=DEV)
    ctx, fs = inp["cond"].to(DEV), inp["fs"].to(DEV)
    with torch.no_grad():
        def run():
            y2 = tiny_unet(x, ts, context=ctx, fs=fs)
            y2b = tiny_unet(x, ts, context=ctx, fs=fs)
            y1 = tiny_unet(x[1:], ts[1:], context=ctx[1:].contiguous(), fs=fs[1:])
            return y2, y2b, y1
        y2, y2b, y1 = _with_backend(hip, run)
    assert torch.equal(y2, y2b), "UNet forward is not deterministic"
    assert torch.equal(y2[1:], y1), "sample output depends on its batch neighbours"


def test_decoder_tiny_vs_reference_golden(hip, tiny_decoder):
    g = load_golden("decoder_tiny.npz")
    refs = [torch.from_numpy(g[f"ref{i}"]).to(DEV) for i in range(5)]
    z = torch.from_numpy(g["z"]).to(DEV)
    with torch.no_grad():
        out = _with_backend(hip, lambda: tiny_decoder.decode_clip(z, refs, scale=1.0 / 0.18215))
        tiny_decoder._ref_cache = None
        out_emu = _with_backend(EmuOps(), lambda: tiny_decoder.decode_clip(z, refs, scale=1.0 / 0.18215))
        tiny_decoder._ref_cache = None
    ref = torch.from_numpy(g["dec_first_stage"])
    e_ref, e_emu = rel_l2(out.cpu(), ref), rel_l2(out.cpu(), out_emu.cpu())
    print(f"tiny decoder: vs reference golden {e_ref:.3e}; vs emulated contract {e_emu:.3e}")
    assert out.shape == ref.shape and torch.isfinite(out).all()
    assert e_ref < 2.0e-2 and e_emu < 2.0e-2        # 1.5 x the measured 1.34e-2 / 1.33e-2


def test_decoder_14_frame_second_pass(hip, tiny_decoder, tiny_sd):
    """inference.py:264-270 decodes a second, shorter clip (frames 1 and T-2 dropped) with the same
    reference context: ragged T and the cached reference K/V must both work.  Checked against the oracle."""
    from oracle import decoder as odec
    g = load_golden("decoder_tiny.npz")
    refs_cpu = [torch.from_numpy(g[f"ref{i}"]) for i in range(5)]
    z = torch.randn(1, 4, 5, 4, 6, generator=torch.Generator().manual_seed(5))
    z2 = z[:, :, [0, 2, 4]]
    dsd = sub_state_dict(tiny_sd, "first_stage_model.decoder.")
    refs = [r.to(DEV) for r in refs_cpu]
    with torch.no_grad():
        a, b = _with_backend(hip, lambda: (tiny_decoder.decode_clip(z.to(DEV), refs, scale=1 / 0.18215),
                                           tiny_decoder.decode_clip(z2.to(DEV), refs, scale=1 / 0.18215)))
        ra = odec.decode_first_stage(dsd, z, refs_cpu)
        rb = odec.decode_first_stage(dsd, z2, refs_cpu)
    ea, eb = rel_l2(a.cpu(), ra), rel_l2(b.cpu(), rb)
    print(f"decoder T=5 vs oracle {ea:.3e}; T=3 (cached refs) vs oracle {eb:.3e}")
    assert ea < 2.4e-2 and eb < 2.0e-2              # 1.5 x the measured 1.60e-2 / 1.31e-2


def _tiny_pipeline(tiny_sd):
    from test_host_logic_cpu import _tiny_model_cfg
    from tooncrafter_amd.utils import instantiate_from_config
    model = instantiate_from_config(dict(target="lvdm.models.ddpm3d.LatentVisualDiffusion",
                                         params=_tiny_model_cfg())).eval()
    model.load_state_dict(tiny_sd, strict=False)
    return model.to(DEV)


def test_ddim_tiny_trajectory_vs_reference_golden(hip, tiny_sd):
    from tooncrafter_amd.lvdm import ddim as my_ddim
    g = load_golden("ddim_tiny.npz")
    model = _tiny_pipeline(tiny_sd)
    noises = torch.from_numpy(g["noises"]).to(DEV)
    dev = lambda k: torch.from_numpy(g[k]).to(DEV)
    cond = {"c_crossattn": [dev("cond")], "c_concat": [dev("c_concat")]}
    uc = {"c_crossattn": [dev("uncond")], "c_concat": [dev("c_concat")]}

    def run():
        it = iter(noises)
        old = my_ddim.noise_like
        my_ddim.noise_like = lambda shape, device, repeat=False: next(it)
        try:
            x0s = []
            s = my_ddim.DDIMSampler(model)
            out, _ = s.sample(S=5, conditioning=cond, batch_size=1, shape=(4, 4, 8, 8), verbose=False,
                              unconditional_guidance_scale=7.5, unconditional_conditioning=uc, eta=1.0, cfg_img=None,
                              mask=None, x0=None, fs=dev("fs"), timestep_spacing="uniform_trailing",
                              guidance_rescale=0.7, x_T=dev("x_T"), unconditional_conditioning_img_nonetext=None,
                              img_callback=lambda p, i: x0s.append(p.clone()))
            return out, x0s
        finally:
            my_ddim.noise_like = old

    with torch.no_grad():
        out, x0s = _with_backend(hip, run)
    errs = [rel_l2(p.cpu(), torch.from_numpy(g["pred_x0"][i])) for i, p in enumerate(x0s)]
    final = rel_l2(out.cpu(), torch.from_numpy(g["samples"]))
    print("DDIM-5 tiny trajectory vs reference: pred_x0 rel-L2 per step", [f"{e:.3e}" for e in errs], f"final {final:.3e}")
    assert torch.isfinite(out).all()
    assert max(errs) < 0.133 and final < 0.108      # 1.5 x the measured 8.84e-2 (worst step) / 7.17e-2


def test_ddim_multicond_tiny_trajectory_vs_reference_golden(hip, tiny_sd):
    """Row f3 on the GPU: samplers/ddim_multiplecond.py mirror, batch-3 UNet pass + fused three-way step."""
    from tooncrafter_amd.lvdm import ddim as my_ddim
    from tooncrafter_amd.lvdm.ddim_multiplecond import DDIMSampler as MultiCondSampler
    g = load_golden("ddim_mc_tiny.npz")
    model = _tiny_pipeline(tiny_sd)
    noises = torch.from_numpy(g["noises"]).to(DEV)
    dev = lambda k: torch.from_numpy(g[k]).to(DEV)
    cond = {"c_crossattn": [dev("cond")], "c_concat": [dev("c_concat")]}
    uc = {"c_crossattn": [dev("uncond")], "c_concat": [dev("c_concat")]}
    uc_img = {"c_crossattn": [dev("uncond_img")], "c_concat": [dev("c_concat")]}

    def run():
        it = iter(noises)
        old = my_ddim.noise_like
        my_ddim.noise_like = lambda shape, device, repeat=False: next(it)
        try:
            x0s = []
            s = MultiCondSampler(model)
            out, _ = s.sample(S=4, conditioning=cond, batch_size=1, shape=(4, 4, 8, 8), verbose=False,
                              unconditional_guidance_scale=7.5, unconditional_conditioning=uc, eta=1.0,
                              cfg_img=float(g["cfg_img"]), mask=None, x0=None, fs=dev("fs"),
                              timestep_spacing="uniform_trailing", guidance_rescale=0.7, x_T=dev("x_T"),
                              unconditional_conditioning_img_nonetext=uc_img,
                              img_callback=lambda p, i: x0s.append(p.clone()))
            return out, x0s
        finally:
            my_ddim.noise_like = old

    with torch.no_grad():
        out, x0s = _with_backend(hip, run)
    errs = [rel_l2(p.cpu(), torch.from_numpy(g["pred_x0"][i])) for i, p in enumerate(x0s)]
    final = rel_l2(out.cpu(), torch.from_numpy(g["samples"]))
    print("multi-cond DDIM-4 tiny trajectory vs reference: pred_x0 rel-L2 per step", [f"{e:.3e}" for e in errs],
          f"final {final:.3e}")
    assert torch.isfinite(out).all()
    assert max(errs) < 0.165 and final < 0.099      # 1.5 x the measured 1.10e-1 (worst step) / 6.59e-2


def test_unet_hipgraph_replay_matches_eager(hip, tiny_unet):
    """The UNet forward is capture-safe (no sync, no host-side allocation outside the caching
    allocator): a hipGraph replay must reproduce the eager result bit for bit."""
    inp = synth.synth_inputs(2, 4, 8, 8, context_dim=96, seed=9)
    x = torch.cat([inp["x_T"], inp["c_concat"]], 1).to(DEV)
    ts = torch.tensor([339, 339], device=DEV)
    ctx, fs = inp["cond"].to(DEV), inp["fs"].to(DEV)

    def run():
        with torch.no_grad():
            eager = tiny_unet(x, ts, context=ctx, fs=fs).clone()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                tiny_unet(x, ts, context=ctx, fs=fs)
            torch.cuda.current_stream().wait_stream(s)
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                y = tiny_unet(x, ts, context=ctx, fs=fs)
            gr.replay()
            torch.cuda.synchronize()
            return eager, y.clone()
    eager, replay = _with_backend(hip, run)
    assert torch.equal(eager, replay)


@pytest.mark.timeout(1500)
def test_unet_full_size_vs_contract(hip, manifest):
    """BASELINE config: 320-channel UNet, 16 frames, 40x64 latent, cond+uncond as B=2.  Checked against
    the PyTorch statement of the operator contract run on the same GPU (the CPU oracle at this size is
    timed -- and compared -- by bench.py's cpu_baseline leg)."""
    from tooncrafter_amd.lvdm.openaimodel3d import UNetModel
    with torch.device("meta"):
        un = UNetModel(**FULL_UNET_CFG)
    un = un.to_empty(device=DEV).eval()
    with torch.no_grad():
        for name, p in un.named_parameters():
            p.copy_(synth.synth_tensor("model.diffusion_model." + name, tuple(p.shape), 1234, DEV))
    inp = synth.synth_inputs(1, 16, 40, 64, seed=7)
    x2 = torch.cat([inp["x_T"]] * 2).to(DEV)
    cc2 = torch.cat([inp["c_concat"]] * 2).to(DEV)
    ctx2 = torch.cat([inp["cond"], inp["uncond"]]).to(DEV)
    ts = torch.tensor([999, 999], device=DEV)
    fs2 = torch.cat([inp["fs"]] * 2).to(DEV)
    with torch.no_grad():
        y = _with_backend(hip, lambda: un(None, ts, context=ctx2, fs=fs2, x_parts=[x2, cc2]))
        torch.cuda.synchronize()
        un._ctx_cache = None
        y_emu = _with_backend(EmuOps(), lambda: un(None, ts, context=ctx2, fs=fs2, x_parts=[x2, cc2]))
    e = rel_l2(y.cpu(), y_emu.cpu())
    print(f"full-size UNet (B=2): HIP vs emulated contract rel-L2 {e:.3e}; out std {float(y.std()):.3f}")
    assert torch.isfinite(y).all()
    assert e < 2.6e-2                                # 1.5 x the measured 1.70e-2


def test_encoder_tiny_vs_reference_golden(hip, tiny_sd):
    """Row f1: encode (Encoder + fused quant_conv) -> posterior + the five hidden states."""
    from tooncrafter_amd.lvdm.autoencoder import AutoencoderKL_Dualref
    g = load_golden("encoder_tiny.npz")
    ae = AutoencoderKL_Dualref(ddconfig=dict(TINY_DD_CFG), embed_dim=4).eval()
    ae.load_state_dict(sub_state_dict(tiny_sd, "first_stage_model."), strict=True)
    ae.to(DEV)
    with torch.no_grad():
        post, hidden = _with_backend(hip, lambda: ae.encode(torch.from_numpy(g["frames"]).to(DEV), return_hidden_states=True))
    z = 0.18215 * post.sample(noise=torch.from_numpy(g["noise"]))
    errs = [rel_l2(h.cpu(), torch.from_numpy(g[f"hid{i}"])) for i, h in enumerate(hidden)]
    ez = rel_l2(z.cpu(), torch.from_numpy(g["z"]))
    print(f"tiny encoder: z rel-L2 {ez:.3e}; hidden states", [f"{e:.3e}" for e in errs])
    assert ez < 3.3e-3 and max(errs) < 1.85e-2      # 1.5 x the measured 2.17e-3 / 1.23e-2


def test_resampler_vs_reference_golden_and_full_size_oracle(hip):
    """Row f2: image-token Resampler on the HIP kernels -- the reference's tiny forward, and the full
    inference_512_v1.0.yaml configuration (4 layers, 12 heads, 256 queries over 257 CLIP tokens) against the
    fp32 CPU oracle on identical synthetic weights."""
    import json, os
    from conftest import GOLDEN as GOLDEN_DIR
    from oracle import resampler as ors
    from tooncrafter_amd import synth
    from tooncrafter_amd.lvdm.resampler import Resampler
    man = json.load(open(os.path.join(GOLDEN_DIR, "resampler_manifest.json")))
    g = load_golden("resampler_tiny.npz")

    def run(cfg, x):
        m = Resampler(**cfg).eval()
        synth.fill_module_(m, prefix="image_proj_model.", seed=1234)
        sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
        m = m.to(DEV)
        with torch.no_grad():
            y = _with_backend(hip, lambda: m(x.to(DEV)))
        return y.cpu(), sd

    y, _ = run(man["tiny_cfg"], torch.from_numpy(g["x"]))
    e_tiny = rel_l2(y, torch.from_numpy(g["y"]))
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(2, 257, 1280, generator=gen)
    y, sd = run(man["full_cfg"], x)
    ref = ors.resampler_forward(sd, x, heads=man["full_cfg"]["heads"])
    e_full = rel_l2(y, ref)
    print(f"resampler: tiny vs reference rel-L2 {e_tiny:.3e}; full config vs CPU oracle rel-L2 {e_full:.3e}")
    assert tuple(y.shape) == (2, 256, 1024) and torch.isfinite(y).all()
    assert e_tiny < 2e-2 and e_full < 2e-2


def test_openclip_towers_vs_transformers_golden(hip):
    """Row f2 against the THIRD-PARTY golden (tests/golden/openclip_hf.npz: HuggingFace transformers' CLIP at the ViT-H/14
    geometry on this repo's synthetic weights, tests/golden/make_openclip_golden.py): the HIP towers of lvdm/openclip.py,
    driven through the reference's own classes (condition.py:215-231 text with layer='penultimate', :340-372 image tokens)."""
    import numpy as np
    from conftest import GOLDEN as GOLDEN_DIR
    from tooncrafter_amd import synth
    from tooncrafter_amd.lvdm.condition import FrozenOpenCLIPEmbedder, FrozenOpenCLIPImageEmbedderV2
    g = np.load(os.path.join(GOLDEN_DIR, "openclip_hf.npz"))
    emb_v = FrozenOpenCLIPImageEmbedderV2().eval()
    synth.fill_module_(emb_v, prefix="embedder.", seed=int(g["seed"]))
    emb_v = emb_v.to(DEV)
    img = torch.from_numpy(g["image"].astype(np.float32)).to(DEV)
    with torch.no_grad():
        yv = _with_backend(hip, lambda: emb_v.model.visual.tokens(img)).cpu()
    del emb_v
    emb_t = FrozenOpenCLIPEmbedder(layer="penultimate").eval()
    synth.fill_module_(emb_t, prefix="cond_stage_model.", seed=int(g["seed"]))
    emb_t = emb_t.to(DEV)
    with torch.no_grad():
        yt = _with_backend(hip, lambda: emb_t(torch.from_numpy(g["tokens"]))).cpu()
    e_v, e_t = rel_l2(yv, torch.from_numpy(g["vision_tokens"])), rel_l2(yt, torch.from_numpy(g["text_tokens"]))
    print(f"openclip towers (HIP) vs transformers golden: ViT-H/14 vision {e_v:.3e} text {e_t:.3e}")
    assert tuple(yv.shape) == (1, 257, 1280) and tuple(yt.shape) == (2, 77, 1024)
    assert e_v < 3e-2 and e_t < 3e-2


def test_openclip_towers_vs_oracle(hip):
    """Row f2 (the restatement itself is pinned to transformers: tests/test_openclip_golden_cpu.py): tiny towers and the full ViT-H/14 geometry -- vision
    32 layers x 16 heads of 80 over 257 tokens, text 23 of 24 layers x 16 heads of 64 over 77 causal tokens --
    against the fp32 CPU restatement on identical synthetic weights."""
    from oracle import openclip as oclip
    from test_host_logic_cpu import _tiny_towers
    from tooncrafter_amd import synth
    from tooncrafter_amd.lvdm.condition import FrozenOpenCLIPEmbedder, FrozenOpenCLIPImageEmbedderV2
    g = torch.Generator().manual_seed(9)

    def on_gpu(mod, fn):
        sd = {k: v.detach().clone() for k, v in mod.state_dict().items()}
        mod = mod.to(DEV)
        with torch.no_grad():
            y = _with_backend(hip, lambda: fn(mod))
        return y.cpu(), sd

    vis, txt = _tiny_towers()
    img, tok = torch.randn(2, 3, 42, 42, generator=g), torch.randint(0, 50, (2, 7), generator=g)
    yv, sdv = on_gpu(vis, lambda m: m.tokens(img.to(DEV)))
    yt, sdt = on_gpu(txt, lambda m: m.tokens(tok.to(DEV), skip_last=1))
    e_tv = rel_l2(yv, oclip.vision_tokens(sdv, img, heads=2))
    e_tt = rel_l2(yt, oclip.text_tokens(sdt, tok, heads=2, skip_last=1))

    emb_v = FrozenOpenCLIPImageEmbedderV2().eval()
    synth.fill_module_(emb_v, prefix="embedder.", seed=1234)
    img = torch.randn(1, 3, 224, 224, generator=g)
    yv, sd = on_gpu(emb_v, lambda m: m.model.visual.tokens(img.to(DEV)))
    sdv = {k[len("model.visual."):]: v for k, v in sd.items() if k.startswith("model.visual.")}
    e_fv = rel_l2(yv, oclip.vision_tokens(sdv, img, heads=16))
    del emb_v
    emb_t = FrozenOpenCLIPEmbedder(layer="penultimate").eval()
    synth.fill_module_(emb_t, prefix="cond_stage_model.", seed=1234)
    tok = torch.randint(0, 49408, (2, 77), generator=g)
    yt, sd = on_gpu(emb_t, lambda m: m(tok))
    sdt = {k[len("model."):]: v for k, v in sd.items() if k.startswith("model.")}
    e_ft = rel_l2(yt, oclip.text_tokens(sdt, tok, heads=16, skip_last=1))
    print(f"openclip towers vs CPU restatement: tiny vision {e_tv:.3e} text {e_tt:.3e}; ViT-H/14 vision {e_fv:.3e} "
          f"text {e_ft:.3e}")
    assert tuple(yv.shape) == (1, 257, 1280) and tuple(yt.shape) == (2, 77, 1024)
    assert max(e_tv, e_tt) < 2e-2 and max(e_fv, e_ft) < 3e-2


def test_pipeline_vs_reference_image_guided_synthesis(hip, tiny_sd):
    """The caller row on the GPU: conditioning -> 2-frame encode -> DDIM (CFG 7.5) -> two decodes -> splice,
    against the reference's own image_guided_synthesis (tests/golden/pipeline_tiny.npz)."""
    import sys
    from conftest import GOLDEN as GOLDEN_DIR
    sys.path.insert(0, GOLDEN_DIR)
    import pipeline_stubs as stubs
    from tooncrafter_amd import clip as pipeline
    from tooncrafter_amd.lvdm import autoencoder as my_ae, ddim as my_ddim
    g = load_golden("pipeline_tiny.npz")
    model = _tiny_pipeline(tiny_sd)
    model.embedder = stubs.StubEmbedder()
    model.image_proj_model = stubs.StubImageProj(4)
    model.get_learned_conditioning = lambda prompts: stubs.stub_text(prompts, DEV)
    videos = torch.from_numpy(g["videos"]).to(DEV)
    noises = torch.from_numpy(g["noises"]).to(DEV)
    torch.manual_seed(2024)
    x_T = torch.randn(1, 4, 4, 8, 8)                 # the CPU draw the golden run used (device generators differ)

    def run():
        it = iter(noises)
        old_noise, old_sample = my_ddim.noise_like, my_ae.DiagonalGaussianDistribution.sample
        my_ddim.noise_like = lambda shape, device, repeat=False: next(it)
        my_ae.DiagonalGaussianDistribution.sample = lambda self, noise=None: self.mean
        try:
            return pipeline.image_guided_synthesis(model, ["ignored"], videos, [1, 4, 4, 8, 8], n_samples=1,
                                                   ddim_steps=3, ddim_eta=1.0, unconditional_guidance_scale=7.5,
                                                   cfg_img=None, fs=10, text_input=False, multiple_cond_cfg=False,
                                                   loop=False, interp=True, timestep_spacing="uniform_trailing",
                                                   guidance_rescale=0.7, x_T=x_T.to(DEV))
        finally:
            my_ddim.noise_like, my_ae.DiagonalGaussianDistribution.sample = old_noise, old_sample

    with torch.no_grad():
        out = _with_backend(hip, run)
    err = rel_l2(out.cpu(), torch.from_numpy(g["out"]))
    print(f"image_guided_synthesis (tiny, 3 steps, CFG 7.5) vs reference: rel-L2 {err:.3e}")
    # three coarse steps at CFG 7.5 followed by the decoder: the bf16 activation noise floor of this tiny run is
    # 0.13 in the emulated contract as well (the exact-arithmetic run of the CPU suite agrees to 1.3e-2)
    assert tuple(out.shape) == (1, 1, 3, 4, 64, 64) and torch.isfinite(out).all() and err < 0.2
