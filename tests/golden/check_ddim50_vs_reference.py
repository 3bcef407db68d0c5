#!/usr/bin/env python3
"""DDIM-50 of the REAL reference sampler against the committed oracle trajectory (build container only: needs
/root/reference; no GPU).

tests/golden/ddim50_oracle.npz is the repo's fp32 oracle (oracle/unet.py + oracle/sampler.py) run in PyTorch-eager on an
MI355X by make_ddim50_golden.py; tests/test_gpu_ddim50.py measures the HIP path against it.  Until round 4 only THREE steps
of that oracle had been laid beside the reference's own sampler (check_fullsize_vs_reference.py).  This script closes the
gap on the metric's own loop: it instantiates the reference's `LatentVisualDiffusion` from the UNMODIFIED
configs/inference_512_v1.0.yaml (1.44 B-parameter UNet; the OpenCLIP conditioners replaced by Identity), fills it with the
same synthetic weights, and runs the reference's `DDIMSampler.sample` (lvdm/models/samplers/ddim.py:60-279:
ddim_sampling / p_sample_ddim, two UNet forwards per step through DiffusionWrapper, ddpm3d.py:1243-1264) for S = 50 steps,
CFG 7.5, guidance rescale 0.7, eta 1, trailing spacing, 16 x 40 x 64 latents, fp32 on the CPU, with the same 50 injected
Gaussian draws (seed 300 + i), and compares every step's pred_x0 and the final latent with the committed file at its
sampled positions.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/check_ddim50_vs_reference.py [steps]

100 full-size UNet forwards: ~70 min on 8 cores.  The per-step table is appended to
profiles/r04_ddim50_oracle_vs_reference.txt as it is produced (a partial run is still a record).
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
OUT = os.path.join(REPO, "profiles", "r04_ddim50_oracle_vs_reference.txt")


def main():
    steps_cap = int(sys.argv[1]) if len(sys.argv) > 1 else None
    mg.install_stubs()
    sys.path.insert(0, REF)
    sys.path.insert(1, REPO)
    sys.path.insert(2, os.path.join(REPO, "tests"))
    import fullsize_cases as fc
    import test_gpu_ddim50 as t50
    from conftest import rel_l2
    from utils.utils import instantiate_from_config            # reference
    from lvdm.models.samplers import ddim as ref_ddim          # reference
    assert ref_ddim.__file__.startswith(REF), ref_ddim.__file__
    torch.set_grad_enabled(False)
    torch.set_num_threads(int(os.environ.get("TC_THREADS", os.cpu_count() or 1)))
    t0 = time.time()

    def log(s):
        line = f"[{time.time() - t0:6.0f}s] {s}"
        print(line, flush=True)

    with open(os.path.join(REF, "configs", "inference_512_v1.0.yaml")) as f:
        cfg = mg.wrap(yaml.safe_load(f))
    p = cfg.model.params
    p.unet_config.params.use_checkpoint = False                # inference.py:286
    p.cond_stage_config = {"target": "torch.nn.Identity"}      # OpenCLIP towers need downloads; not on this path
    p.img_cond_stage_config = {"target": "torch.nn.Identity"}
    p.image_proj_stage_config = {"target": "torch.nn.Identity"}
    torch.manual_seed(0)
    model = instantiate_from_config(cfg.model).eval()
    sd = fc.full_state_dict(("model.diffusion_model.",))
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(not k.startswith("model.diffusion_model.") for k in missing)
    del sd
    model.first_stage_model = None                             # not on this path; frees 0.4 GB
    log("reference LatentVisualDiffusion from the unmodified YAML, weights = synth seed 1234")

    g = np.load(t50.GOLDEN)
    S = t50.S
    gold_x0, gold_final = torch.from_numpy(g["x0_fp32"]), torch.from_numpy(g["final_fp32"])
    floor_x0, floor_final = g["floor_x0"].tolist(), float(g["floor_final"])
    inp = fc.inputs()
    ix0, ifin, _ = t50.sample_positions(inp["x_T"].numel(), inp["x_T"].numel(), 1)
    noises = [torch.randn(1, 4, fc.T, fc.H, fc.W, generator=torch.Generator().manual_seed(300 + i)) for i in range(S)]

    header = [f"# real reference DDIMSampler (/root/reference/lvdm/models/samplers/ddim.py, unmodified inference_512_v1.0.yaml, fp32 CPU, "
              f"torch {torch.__version__}) vs tests/golden/ddim50_oracle.npz ({str(g['meta'])})",
              f"# DDIM-{S} CFG {fc.CFG} eta {fc.ETA} rescale {fc.RESCALE}, 16x40x64 latents, injected noise seed 300+i; rel-L2 over the "
              f"{t50.N_X0} sampled positions of each step's pred_x0; floor = the bf16-autocast oracle's distance from the fp32 oracle",
              "step  reference-vs-oracle  bf16 floor  ratio"]
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        f.write("\n".join(header) + "\n")
    print("\n".join(header), flush=True)

    ref_ddim.DDIMSampler.register_buffer = lambda self, n, a: setattr(self, n, a)     # ddim.py:18-22 hard-codes cuda
    it = iter(noises)
    ref_ddim.noise_like = lambda shape, device, repeat=False: next(it)
    cond = {"c_crossattn": [inp["cond"]], "c_concat": [inp["c_concat"]]}
    uc = {"c_crossattn": [inp["uncond"]], "c_concat": [inp["c_concat"]]}
    errs = []

    class Stop(Exception):
        pass

    def on_step(pred_x0, i):
        k = len(errs)
        e = rel_l2(pred_x0.reshape(-1)[ix0], gold_x0[k])
        errs.append(e)
        row = f"{k:4d}  {e:.3e}  {floor_x0[k]:.3e}  {e / floor_x0[k]:.2e}"
        log(row)
        with open(OUT, "a") as f:
            f.write(row + "\n")
        if steps_cap is not None and len(errs) >= steps_cap:
            raise Stop

    final = None
    try:
        final, _ = ref_ddim.DDIMSampler(model).sample(
            S=S, conditioning=cond, batch_size=1, shape=(4, fc.T, fc.H, fc.W), verbose=False,
            unconditional_guidance_scale=fc.CFG, unconditional_conditioning=uc, eta=fc.ETA, cfg_img=None, mask=None,
            x0=None, fs=inp["fs"], timestep_spacing="uniform_trailing", guidance_rescale=fc.RESCALE, x_T=inp["x_T"],
            unconditional_conditioning_img_nonetext=None, img_callback=on_step)
    except Stop:
        pass
    tail = []
    if final is not None:
        ef = rel_l2(final.reshape(-1)[ifin], gold_final)
        tail.append(f"final latent: reference-vs-oracle {ef:.3e}  bf16 floor {floor_final:.3e}  ratio {ef / floor_final:.2e}")
        errs.append(ef)
    tail.append(f"worst {max(errs):.3e} over {len(errs)} comparisons; wall {time.time() - t0:.0f} s on {torch.get_num_threads()} threads")
    with open(OUT, "a") as f:
        f.write("\n".join(tail) + "\n")
    print("\n".join(tail))
    # the oracle was run in fp32 on a GPU (rocBLAS / MIOpen summation orders), the reference here on CPU kernels: fp32
    # rounding differences, amplified by CFG 7.5 along the trajectory exactly like the bf16 floor is; two orders below it
    assert max(errs) < 2e-3, max(errs)


if __name__ == "__main__":
    main()
