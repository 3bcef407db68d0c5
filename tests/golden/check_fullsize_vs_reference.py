#!/usr/bin/env python3
"""Cross-check of the FULL-SIZE oracle goldens against the REAL reference (build container only: needs /root/reference).

tests/golden/fullsize_oracle.npz is produced by the repo's own fp32 oracle (make_fullsize_golden.py); the oracle is pinned
to the reference at tiny widths (make_golden.py).  This script closes the remaining gap: it instantiates the reference's
own `LatentVisualDiffusion` from the UNMODIFIED configs/inference_512_v1.0.yaml (1.44 B-parameter UNet, ch = 128
VideoDecoder; only the two OpenCLIP conditioners are replaced by Identity -- they need downloaded weights), fills it with
the same synthetic weights, runs the same full-size cases and compares with the committed file:

  unet    one UNetModel forward (B = 1, t = 601, 16 x 40 x 64 latents)                     openaimodel3d.py:548-603
  dec     decode_first_stage with ref_context, 16 frames and the 14-frame re-decode          ddpm3d.py:647-679,
          (perframe_ae = True as inference.py:289 sets it), sampled outputs + global norm    autoencoder_dualref.py:489-527
  ddim    3-step DDIMSampler.sample, CFG 7.5, rescale 0.7, eta 1, trailing, injected noise   samplers/ddim.py:60-279

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/check_fullsize_vs_reference.py [unet] [dec] [ddim]

~10 min on 8 cores (UNet forward 34 s, decoder 90 s, DDIM 6 forwards).  Expected rel-L2 ~1e-6 (fp32 summation order only).
The table is written to profiles/r03_fullsize_golden_vs_reference.txt.
"""
import os
import sys
import time

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.dont_write_bytecode = True
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (import stubs + AttrDict, nothing from the reference yet)

REF = mg.REF


def main():
    only = set(sys.argv[1:])
    want = lambda k: not only or k in only
    mg.install_stubs()
    sys.path.insert(0, REF)
    sys.path.insert(1, REPO)
    sys.path.insert(2, os.path.join(REPO, "tests"))
    import fullsize_cases as fc
    from conftest import rel_l2
    from utils.utils import instantiate_from_config            # reference
    from lvdm.models.samplers import ddim as ref_ddim          # reference
    assert ref_ddim.__file__.startswith(REF), ref_ddim.__file__
    torch.set_grad_enabled(False)
    torch.set_num_threads(os.cpu_count() or 1)
    t0 = time.time()
    log = lambda s: print(f"[{time.time() - t0:6.0f}s] {s}", flush=True)

    with open(os.path.join(REF, "configs", "inference_512_v1.0.yaml")) as f:
        cfg = mg.wrap(yaml.safe_load(f))
    p = cfg.model.params
    p.unet_config.params.use_checkpoint = False                # inference.py:286
    p.cond_stage_config = {"target": "torch.nn.Identity"}      # OpenCLIP towers need downloads; not on this path
    p.img_cond_stage_config = {"target": "torch.nn.Identity"}
    p.image_proj_stage_config = {"target": "torch.nn.Identity"}
    torch.manual_seed(0)
    model = instantiate_from_config(cfg.model).eval()
    model.perframe_ae = True                                   # inference.py:289
    sd = fc.full_state_dict(("model.diffusion_model.", "first_stage_model.decoder."))
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(not k.startswith(("model.diffusion_model.", "first_stage_model.decoder.")) for k in missing)
    del sd
    log(f"reference LatentVisualDiffusion from the unmodified YAML, {sum(p.numel() for p in model.parameters()) / 1e9:.3f} B parameters")
    golden = dict(np.load(fc.GOLDEN_FILE))
    inp = fc.inputs()
    rows = []

    def row(name, got, ref):
        e = rel_l2(got, torch.from_numpy(np.asarray(ref)))
        rows.append(f"{name:34s} rel-L2 {e:.3e}")
        log(rows[-1])
        return e

    worst = 0.0
    if want("unet"):
        y = model.model.diffusion_model(torch.cat([inp["x_T"], inp["c_concat"]], 1), torch.tensor([fc.UNET_T]),
                                        context=inp["cond"], fs=inp["fs"])
        worst = max(worst, row("UNet forward (whole tensor)", y, golden["unet_y"]))

    if want("dec"):
        for tag, z in (("dec16", inp["z_dec"]), ("dec14", inp["z_dec"][:, :, fc.IDX14].contiguous())):
            y = model.decode_first_stage(z, ref_context=inp["refs"])
            flat = y.reshape(-1)
            worst = max(worst, row(f"decoder {tag} (131072 samples)", flat[fc.sample_idx(flat.numel(), fc.N_OUT, 1)],
                                   golden[f"{tag}_out"]))
            nr = float(y.double().norm()) / float(golden[f"{tag}_out_norm"])
            rows.append(f"decoder {tag} |y| / |golden|          {nr:.8f}")
            log(rows[-1])
            worst = max(worst, abs(nr - 1.0))

    if want("ddim"):
        ref_ddim.DDIMSampler.register_buffer = lambda self, n, a: setattr(self, n, a)     # ddim.py:18-22 hard-codes cuda
        it = iter(inp["noises"])
        ref_ddim.noise_like = lambda shape, device, repeat=False: next(it)
        cond = {"c_crossattn": [inp["cond"]], "c_concat": [inp["c_concat"]]}
        uc = {"c_crossattn": [inp["uncond"]], "c_concat": [inp["c_concat"]]}
        x0s = []
        out, _ = ref_ddim.DDIMSampler(model).sample(
            S=fc.DDIM_STEPS, conditioning=cond, batch_size=1, shape=(4, fc.T, fc.H, fc.W), verbose=False,
            unconditional_guidance_scale=fc.CFG, unconditional_conditioning=uc, eta=fc.ETA, cfg_img=None, mask=None,
            x0=None, fs=inp["fs"], timestep_spacing="uniform_trailing", guidance_rescale=fc.RESCALE, x_T=inp["x_T"],
            unconditional_conditioning_img_nonetext=None, img_callback=lambda p, i: x0s.append(p.clone()))
        for i, pr in enumerate(x0s):
            worst = max(worst, row(f"DDIM-3 pred_x0 step {i}", pr, golden[f"ddim_pred_x0_{i}"]))
        worst = max(worst, row("DDIM-3 final latent", out, golden["ddim_final"]))

    text = "\n".join(["# real reference (/root/reference, unmodified inference_512_v1.0.yaml, fp32 CPU, torch "
                      f"{torch.__version__}) vs tests/golden/fullsize_oracle.npz (repo oracle)", *rows,
                      f"worst {worst:.3e}"])
    if not only:
        os.makedirs(os.path.join(REPO, "profiles"), exist_ok=True)
        with open(os.path.join(REPO, "profiles", "r03_fullsize_golden_vs_reference.txt"), "w") as f:
            f.write(text + "\n")
    print(text)
    assert worst < 1e-4, worst


if __name__ == "__main__":
    main()
