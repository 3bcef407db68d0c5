#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REAL reference.

Run in the build container only (the GPU box has no /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference tree is imported read-only with import stubs for the four
third-party modules this image lacks (cv2, pytorch_lightning, torchvision,
xformers) -- the recipe of SURVEY.md Appendix A.  Nothing is copied from it; the
script only calls its public classes and saves input/output tensors:

  schedule.npz       DDPM.register_schedule buffers + DDIMSampler.make_schedule tables
                     for the full 1000-step config (S=50 eta=1 trailing, S=2, S=50 uniform)
  unet_tiny.npz      UNetModel forward, 64-channel config, T=4, 8x8 latent
  decoder_tiny.npz   VideoDecoder forward via AutoencoderKL_Dualref.decode, ch=64, T=3
  ddim_tiny.npz      5-step DDIM trajectory (CFG 7.5, rescale 0.7, eta=1, injected noise)
  pipeline_tiny.npz  scripts/evaluation/inference.py::image_guided_synthesis on the tiny model with the deterministic
                     conditioner stand-ins of pipeline_stubs.py (interp mode, 3 DDIM steps, CFG 7.5) -- the caller row
  resampler_tiny.npz image-token Resampler forward (2 layers, dim 128, 2 heads of 64, 4 queries x 3 frames) -- row f2
  ddim_mc_tiny.npz   4-step trajectory of samplers/ddim_multiplecond.py (three-way guidance: text 7.5,
                     image 3.0, rescale 0.7, eta=1, injected noise) -- SURVEY.md row f3
                     through LatentVisualDiffusion.apply_model, then decode_first_stage

Weights are the deterministic synthetic recipe of tooncrafter_amd/synth.py, keyed
by the reference's own parameter names, so any implementation can rebuild them.
"""
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = os.environ.get("TOONCRAFTER_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    _mod("cv2")

    class LightningModule(nn.Module):
        @property
        def device(self):
            try:
                return next(self.parameters()).device
            except StopIteration:
                return torch.device("cpu")

    pl = _mod("pytorch_lightning", LightningModule=LightningModule, seed_everything=torch.manual_seed)
    pl.utilities = _mod("pytorch_lightning.utilities", rank_zero_only=lambda f: f)
    tv = _mod("torchvision")
    tv.utils = _mod("torchvision.utils", make_grid=lambda *a, **k: None)
    _mod("torchvision.transforms")

    def mea(q, k, v, attn_bias=None, op=None):
        outs = [F.scaled_dot_product_attention(q[i:i + 8, None], k[i:i + 8, None], v[i:i + 8, None])[:, 0]
                for i in range(0, q.shape[0], 8)]
        return torch.cat(outs, 0)

    xf = _mod("xformers", __version__="0.0.20")
    xf.ops = _mod("xformers.ops", memory_efficient_attention=mea)


class AttrDict(dict):
    """Stand-in for OmegaConf's DictConfig: attribute access + dict protocol."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def wrap(o):
    if isinstance(o, dict):
        return AttrDict({k: wrap(v) for k, v in o.items()})
    if isinstance(o, list):
        return [wrap(v) for v in o]
    return o


def tiny_config():
    """The reference's inference YAML with the widths shrunk; structure untouched."""
    with open(os.path.join(REF, "configs", "inference_512_v1.0.yaml")) as f:
        cfg = yaml.safe_load(f)
    p = cfg["model"]["params"]
    p["image_size"] = [8, 8]
    u = p["unet_config"]["params"]
    u["model_channels"] = 64
    u["context_dim"] = 96
    u["temporal_length"] = 4
    u["use_checkpoint"] = False
    d = p["first_stage_config"]["params"]["ddconfig"]
    d["ch"] = 64
    p["cond_stage_config"] = {"target": "torch.nn.Identity"}
    p["img_cond_stage_config"] = {"target": "torch.nn.Identity"}
    p["image_proj_stage_config"] = {"target": "torch.nn.Identity"}
    return wrap(cfg)


def main():
    install_stubs()
    sys.path.insert(0, REF)
    sys.path.insert(1, REPO)
    from tooncrafter_amd import synth                        # noqa: E402  (ours: weight recipe only)
    from utils.utils import instantiate_from_config          # noqa: E402  (reference)
    from lvdm.models.samplers import ddim as ref_ddim        # noqa: E402
    from lvdm.models.ddpm3d import DDPM                      # noqa: E402
    assert ref_ddim.__file__.startswith(REF), ref_ddim.__file__

    torch.manual_seed(0)
    torch.set_grad_enabled(False)

    # ---------------------------------------------------------------- schedule (full config)
    class Stub(nn.Module):
        rescale_betas_zero_snr = True
        parameterization = "v"
        v_posterior = 0.0
        use_dynamic_rescale = True
        device = torch.device("cpu")

    stub = Stub()
    DDPM.register_schedule(stub, beta_schedule="linear", timesteps=1000,
                           linear_start=0.00085, linear_end=0.012)
    ref_ddim.DDIMSampler.register_buffer = lambda self, n, a: setattr(self, n, a)   # CPU (ddim.py:18-22 hard-codes cuda)
    out = {k: getattr(stub, k).numpy() for k in
           ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
            "sqrt_one_minus_alphas_cumprod")}
    # scale_arr comes from LatentDiffusion.__init__; taken from the tiny model below (same code path).

    cfg = tiny_config()
    model = instantiate_from_config(cfg.model).eval()
    model.perframe_ae = True
    synth.fill_module_(model, seed=1234)
    out["scale_arr"] = model.scale_arr.numpy()
    stub.scale_arr = model.scale_arr
    for tag, S, eta, method in (("s50_trailing", 50, 1.0, "uniform_trailing"),
                                ("s2_trailing", 2, 1.0, "uniform_trailing"),
                                ("s50_uniform", 50, 0.0, "uniform"),
                                ("s5_trailing", 5, 1.0, "uniform_trailing")):
        s = ref_ddim.DDIMSampler(stub)
        s.make_schedule(S, ddim_discretize=method, ddim_eta=eta, verbose=False)
        out[tag + "_timesteps"] = np.asarray(s.ddim_timesteps)
        out[tag + "_alphas"] = np.asarray(s.ddim_alphas, dtype=np.float64)
        out[tag + "_alphas_prev"] = np.asarray(s.ddim_alphas_prev, dtype=np.float64)
        out[tag + "_sigmas"] = np.asarray(s.ddim_sigmas, dtype=np.float64)
        out[tag + "_scale_arr"] = s.ddim_scale_arr.numpy()
        out[tag + "_scale_arr_prev"] = s.ddim_scale_arr_prev.numpy()
        # the fp32 scalars p_sample_ddim actually materialises with torch.full (ddim.py:251-265)
        out[tag + "_radicand_f32"] = np.asarray(
            [float((1. - torch.full((1,), s.ddim_alphas_prev[i]) - torch.full((1,), s.ddim_sigmas[i]) ** 2)[0])
             for i in range(S)], dtype=np.float32)
    np.savez_compressed(os.path.join(HERE, "schedule.npz"), **out)
    print("schedule.npz written")

    # ---------------------------------------------------------------- parameter manifests
    def manifest(m):
        return {k: list(v.shape) for k, v in m.named_parameters()}

    man = {"tiny": manifest(model)}
    with open(os.path.join(REF, "configs", "inference_512_v1.0.yaml")) as f:
        full = wrap(yaml.safe_load(f))
    fp = full.model.params
    fp.unet_config.params.use_checkpoint = False
    with torch.device("meta"):
        from lvdm.modules.networks.openaimodel3d import UNetModel
        from lvdm.models.autoencoder_dualref import VideoDecoder
        from lvdm.modules.networks.ae_modules import Encoder
        un = UNetModel(**fp.unet_config.params)
        vd = VideoDecoder(**fp.first_stage_config.params.ddconfig)
        en = Encoder(**fp.first_stage_config.params.ddconfig)
    man["full"] = {**{"model.diffusion_model." + k: v for k, v in manifest(un).items()},
                   **{"first_stage_model.decoder." + k: v for k, v in manifest(vd).items()},
                   **{"first_stage_model.encoder." + k: v for k, v in manifest(en).items()},
                   "first_stage_model.quant_conv.weight": [8, 8, 1, 1], "first_stage_model.quant_conv.bias": [8],
                   "first_stage_model.post_quant_conv.weight": [4, 4, 1, 1],
                   "first_stage_model.post_quant_conv.bias": [4]}
    man["full_buffers"] = {k: list(v.shape) for k, v in model.named_buffers()
                           if not k.startswith(("first_stage_model", "model."))}
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(man, f, separators=(",", ":"), sort_keys=True)
    print("manifest.json written:", {k: len(v) for k, v in man.items()})

    # ---------------------------------------------------------------- tiny UNet forward
    T, H, W = 4, 8, 8
    inp = synth.synth_inputs(1, T, H, W, context_dim=96, seed=7)
    unet = model.model.diffusion_model
    x = torch.cat([inp["x_T"], inp["c_concat"]], dim=1)
    ts = torch.tensor([601], dtype=torch.long)
    y = unet(x, ts, context=inp["cond"], fs=inp["fs"])
    n_params = sum(p.numel() for p in unet.parameters())
    np.savez_compressed(os.path.join(HERE, "unet_tiny.npz"), x=x.numpy(), timesteps=ts.numpy(),
                        context=inp["cond"].numpy(), fs=inp["fs"].numpy(), y=y.numpy(),
                        n_params=np.int64(n_params))
    print("unet_tiny.npz written; out std", float(y.std()), "params", n_params)

    # ---------------------------------------------------------------- tiny decoder forward
    Td, hd, wd = 3, 4, 6
    g = torch.Generator().manual_seed(21)
    z = torch.randn(1, 4, Td, hd, wd, generator=g)
    ref_ctx = synth.synth_ref_context(1, hd, wd, ch=64, seed=11)
    dec_in = (1.0 / 0.18215) * z.permute(0, 2, 1, 3, 4).reshape(Td, 4, hd, wd)
    dec = model.first_stage_model.decode(dec_in, ref_context=ref_ctx, timesteps=Td)
    # and through the pipeline-level entry point (z * 1/scale_factor inside)
    model.temporal_length = Td
    dec2 = model.decode_first_stage(z, ref_context=ref_ctx)
    model.temporal_length = cfg.model.params.unet_config.params.temporal_length
    nd = sum(p.numel() for p in model.first_stage_model.decoder.parameters())
    np.savez_compressed(os.path.join(HERE, "decoder_tiny.npz"), z=z.numpy(),
                        **{f"ref{i}": r.numpy() for i, r in enumerate(ref_ctx)},
                        dec=dec.numpy(), dec_first_stage=dec2.numpy(), n_params=np.int64(nd))
    print("decoder_tiny.npz written; out std", float(dec.std()), "params", nd)

    # ---------------------------------------------------------------- tiny encoder (row f1)
    ge = torch.Generator().manual_seed(31)
    frames = torch.randn(3, 3, 32, 48, generator=ge).clamp(-1, 1)
    post, hidden = model.first_stage_model.encode(frames, return_hidden_states=True)
    enoise = torch.randn(post.mean.shape, generator=ge)
    zs = model.scale_factor * post.sample(noise=enoise)      # get_first_stage_encoding (ddpm3d.py:610-618) with injected noise
    ne = sum(p.numel() for p in model.first_stage_model.encoder.parameters())
    np.savez_compressed(os.path.join(HERE, "encoder_tiny.npz"), frames=frames.numpy(), noise=enoise.numpy(),
                        mean=post.mean.numpy(), logvar=post.logvar.numpy(), z=zs.numpy(),
                        **{f"hid{i}": h.numpy() for i, h in enumerate(hidden)}, n_params=np.int64(ne))
    print("encoder_tiny.npz written; z std", float(zs.std()), "params", ne)

    # ---------------------------------------------------------------- tiny DDIM trajectory
    S = 5
    ng = torch.Generator().manual_seed(99)
    noises = [torch.randn(inp["x_T"].shape, generator=ng) for _ in range(S)]
    it = iter(noises)
    ref_ddim.noise_like = lambda shape, device, repeat=False: next(it)    # name imported at ddim.py:5
    sampler = ref_ddim.DDIMSampler(model)
    cond = {"c_crossattn": [inp["cond"]], "c_concat": [inp["c_concat"]]}
    uc = {"c_crossattn": [inp["uncond"]], "c_concat": [inp["c_concat"]]}
    x0s = []
    samples, _ = sampler.sample(S=S, conditioning=cond, batch_size=1, shape=(4, T, H, W), verbose=False,
                                unconditional_guidance_scale=7.5, unconditional_conditioning=uc, eta=1.0,
                                cfg_img=None, mask=None, x0=None, fs=inp["fs"],
                                timestep_spacing="uniform_trailing", guidance_rescale=0.7, x_T=inp["x_T"],
                                unconditional_conditioning_img_nonetext=None,
                                img_callback=lambda p, i: x0s.append(p.clone()))
    np.savez_compressed(os.path.join(HERE, "ddim_tiny.npz"), x_T=inp["x_T"].numpy(),
                        c_concat=inp["c_concat"].numpy(), cond=inp["cond"].numpy(),
                        uncond=inp["uncond"].numpy(), fs=inp["fs"].numpy(),
                        noises=torch.stack(noises).numpy(), pred_x0=torch.stack(x0s).numpy(),
                        samples=samples.numpy())
    print("ddim_tiny.npz written; final std", float(samples.std()), "finite", bool(torch.isfinite(samples).all()))

    # ---------------------------------------------------------------- tiny multi-condition DDIM trajectory (f3)
    from lvdm.models.samplers import ddim_multiplecond as ref_mc            # noqa: E402
    assert ref_mc.__file__.startswith(REF), ref_mc.__file__
    ref_mc.DDIMSampler.register_buffer = lambda self, n, a: setattr(self, n, a)
    S = 4
    ng = torch.Generator().manual_seed(123)
    noises = [torch.randn(inp["x_T"].shape, generator=ng) for _ in range(S)]
    it2 = iter(noises)
    ref_mc.noise_like = lambda shape, device, repeat=False: next(it2)
    # third condition: text dropped, image tokens kept (what funcs.py would build for img-only guidance)
    n_text = 77
    uc_img_ctx = torch.cat([inp["uncond"][:, :n_text], inp["cond"][:, n_text:]], dim=1)
    uc_img = {"c_crossattn": [uc_img_ctx], "c_concat": [inp["c_concat"]]}
    sampler = ref_mc.DDIMSampler(model)
    x0s = []
    samples, _ = sampler.sample(S=S, conditioning=cond, batch_size=1, shape=(4, T, H, W), verbose=False,
                                unconditional_guidance_scale=7.5, unconditional_conditioning=uc, eta=1.0,
                                cfg_img=3.0, mask=None, x0=None, fs=inp["fs"],
                                timestep_spacing="uniform_trailing", guidance_rescale=0.7, x_T=inp["x_T"],
                                unconditional_conditioning_img_nonetext=uc_img,
                                img_callback=lambda p, i: x0s.append(p.clone()))
    np.savez_compressed(os.path.join(HERE, "ddim_mc_tiny.npz"), x_T=inp["x_T"].numpy(),
                        c_concat=inp["c_concat"].numpy(), cond=inp["cond"].numpy(),
                        uncond=inp["uncond"].numpy(), uncond_img=uc_img_ctx.numpy(), fs=inp["fs"].numpy(),
                        noises=torch.stack(noises).numpy(), pred_x0=torch.stack(x0s).numpy(),
                        samples=samples.numpy(), cfg_img=np.float32(3.0))
    print("ddim_mc_tiny.npz written; final std", float(samples.std()), "finite", bool(torch.isfinite(samples).all()))


def pipeline_golden():
    """The caller of the hot path: the reference's own image_guided_synthesis (inference.py:180-277) end to end."""
    import importlib.util
    install_stubs()
    _mod("omegaconf", OmegaConf=type("OmegaConf", (), {}))
    for pth in (REF, REPO, HERE):
        if pth not in sys.path:
            sys.path.insert(0, pth)
    sys.path.insert(0, REF)
    from tooncrafter_amd import synth                        # noqa: E402  (ours: weight recipe only)
    from utils.utils import instantiate_from_config          # noqa: E402  (reference)
    from lvdm.models.samplers import ddim as ref_ddim        # noqa: E402
    from lvdm import distributions as ref_dist               # noqa: E402
    import pipeline_stubs as stubs                           # noqa: E402
    assert ref_ddim.__file__.startswith(REF)
    spec = importlib.util.spec_from_file_location("ref_inference", os.path.join(REF, "scripts/evaluation/inference.py"))
    ref_inf = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_inf)
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    cfg = tiny_config()
    model = instantiate_from_config(cfg.model).eval()
    model.perframe_ae = True
    synth.fill_module_(model, seed=1234)
    T = cfg.model.params.unet_config.params.temporal_length
    model.embedder = stubs.StubEmbedder()
    model.image_proj_model = stubs.StubImageProj(T)
    model.get_learned_conditioning = lambda prompts: stubs.stub_text(prompts)
    ref_dist.DiagonalGaussianDistribution.sample = lambda self, noise=None: self.mean       # deterministic posterior
    ref_ddim.DDIMSampler.register_buffer = lambda self, n, a: setattr(self, n, a)
    g = torch.Generator().manual_seed(61)
    fa, fb = (torch.randn(1, 3, 1, 64, 64, generator=g).clamp(-1, 1) for _ in range(2))
    videos = torch.cat([fa.repeat(1, 1, T // 2, 1, 1), fb.repeat(1, 1, T // 2, 1, 1)], dim=2)   # load_data_prompts interp
    S = 3
    noises = [torch.randn(1, 4, T, 8, 8, generator=g) for _ in range(S)]
    it = iter(noises)
    ref_ddim.noise_like = lambda shape, device, repeat=False: next(it)
    torch.manual_seed(2024)                                   # x_T = the first torch.randn of ddim_sampling
    out = ref_inf.image_guided_synthesis(model, ["ignored"], videos, [1, 4, T, 8, 8], n_samples=1, ddim_steps=S,
                                         ddim_eta=1.0, unconditional_guidance_scale=7.5, cfg_img=None, fs=10,
                                         text_input=False, multiple_cond_cfg=False, loop=False, interp=True,
                                         timestep_spacing="uniform_trailing", guidance_rescale=0.7)
    z, hs = ref_inf.get_latent_z_with_hidden_states(model, videos)
    np.savez_compressed(os.path.join(HERE, "pipeline_tiny.npz"), videos=videos.numpy(),
                        noises=torch.stack(noises).numpy(), out=out.numpy(), z=z.numpy(),
                        hs_shapes=np.asarray([list(h.shape) for h in hs]),
                        **{f"hs{i}": h[:, ::4, :, ::4, ::4].numpy() for i, h in enumerate(hs)})      # subsampled: small fixture
    print("pipeline_tiny.npz written; out", tuple(out.shape), "std", float(out.std()), "finite", bool(torch.isfinite(out).all()))


def resampler_golden():
    """Row f2: the reference Resampler (lvdm/modules/encoders/resampler.py) at a tiny width, plus the parameter
    manifest of the full inference_512_v1.0.yaml configuration."""
    import importlib.util
    if REPO not in sys.path:
        sys.path.insert(0, REPO)
    from tooncrafter_amd import synth                        # noqa: E402  (ours: weight recipe only)
    # the file imports nothing but torch/math: load it by path (the package __init__ chain is not needed)
    spec = importlib.util.spec_from_file_location("ref_resampler", os.path.join(REF, "lvdm/modules/encoders/resampler.py"))
    ref_rs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_rs)
    assert ref_rs.__file__.startswith(REF), ref_rs.__file__
    tiny = dict(dim=128, depth=2, dim_head=64, heads=2, num_queries=4, embedding_dim=96, output_dim=64, ff_mult=4,
                video_length=3)
    torch.manual_seed(0)
    m = ref_rs.Resampler(**tiny).eval()
    synth.fill_module_(m, prefix="image_proj_model.", seed=1234)
    g = torch.Generator().manual_seed(41)
    x = torch.randn(2, 9, 96, generator=g)
    with torch.no_grad():
        y = m(x)
    # ImageProjModel (resampler.py:9-23; unused by inference_512_v1.0.yaml, kept for checkpoints that use it)
    ip = ref_rs.ImageProjModel(cross_attention_dim=64, clip_embeddings_dim=96, clip_extra_context_tokens=4).eval()
    synth.fill_module_(ip, prefix="image_proj_model.", seed=1234)
    xe = torch.randn(3, 96, generator=g)
    with torch.no_grad():
        ye = ip(xe)
    np.savez_compressed(os.path.join(HERE, "resampler_tiny.npz"), x=x.numpy(), y=y.numpy(),
                        n_params=np.int64(sum(p.numel() for p in m.parameters())),
                        proj_x=xe.numpy(), proj_y=ye.numpy())
    full = dict(dim=1024, depth=4, dim_head=64, heads=12, num_queries=16, embedding_dim=1280, output_dim=1024,
                ff_mult=4, video_length=16)
    with torch.device("meta"):
        mf = ref_rs.Resampler(**full)
    man = {"tiny_cfg": tiny, "full_cfg": full,
           "tiny": {k: list(v.shape) for k, v in m.state_dict().items()},
           "full": {k: list(v.shape) for k, v in mf.state_dict().items()}}
    with open(os.path.join(HERE, "resampler_manifest.json"), "w") as f:
        json.dump(man, f, indent=0, sort_keys=True)
    print("resampler_tiny.npz written; out std", float(y.std()), "params", sum(p.numel() for p in m.parameters()))


if __name__ == "__main__":
    if sys.argv[1:] == ["resampler"]:          # regenerate only the row-f2 fixtures
        resampler_golden()
    elif sys.argv[1:] == ["pipeline"]:         # regenerate only the caller-row fixture
        pipeline_golden()
    else:
        main()
        resampler_golden()
        print("run `python make_golden.py pipeline` separately (it patches the reference's posterior sampling)")
