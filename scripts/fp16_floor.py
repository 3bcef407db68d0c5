#!/usr/bin/env python3
"""The parity floor of the reference's OWN precision (VERDICT r5, weak #1 / missing #6): the reference samples under
`torch.autocast(float16)` (scripts/evaluation/inference.py:323), while every tolerance of this repo is calibrated against
the oracle under bf16 autocast (scripts/noise_floor.py, tests/golden/ddim50_oracle.npz) because bf16 is what BASELINE.json
configs[1] asks the HIP path to compute in.  This script measures the fp16 floor the same way the bf16 one was measured --
the ORACLE code (oracle/unet.py, oracle/sampler.py, oracle/decoder.py: PyTorch-eager restatements of the reference) on this
GPU under `torch.autocast(float16)` against the same oracle in fp32 -- for one UNet forward, one 16-frame decode and the
full DDIM-50 / CFG 7.5 trajectory at the BASELINE shape, and prints it BESIDE the bf16 floor and the HIP path's own
distance (the latter two from tests/golden/ddim50_oracle.npz and, if present, gpurun_out/ddim50_parity_bf16.txt).

The fp32 trajectory is the committed golden's (sampled positions: tests/test_gpu_ddim50.py `sample_positions`), so only
the fp16 trajectory is run here (~90 s); `--live-fp32` re-runs the fp32 oracle as well and compares full tensors.

Test infrastructure: imports oracle/ and tests/, never imported by the product.
    python scripts/fp16_floor.py > gpurun_out/fp16_floor.txt
"""
import argparse
import os
import re
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import fullsize_cases as fc  # noqa: E402
import test_gpu_ddim50 as t50  # noqa: E402
from conftest import rel_l2, sub_state_dict  # noqa: E402
from oracle import decoder as odec  # noqa: E402
from oracle import sampler as osamp  # noqa: E402
from oracle import unet as ounet  # noqa: E402

DEV = "cuda"


def build_weights():
    import bench
    from tooncrafter_amd import synth
    from tooncrafter_amd.utils import instantiate_from_config
    with torch.device("meta"):
        model = instantiate_from_config(dict(target="lvdm.models.ddpm3d.LatentVisualDiffusion", params=bench.MODEL_PARAMS))
    sd = {}
    for name, p in model.named_parameters():
        if name.startswith("model.diffusion_model.") or name.startswith("first_stage_model.decoder."):
            sd[name] = synth.synth_tensor(name, tuple(p.shape), 1234, "cpu").to(DEV)
    return sd


def trajectory(usd, dsd, inp, dtype, steps):
    """The oracle's DDIM trajectory + decode under autocast(dtype) (dtype None = fp32), full tensors."""
    dev = lambda k: inp[k].to(DEV)
    noises = t50._noises()
    sched = osamp.make_schedule_buffers()
    cc = dev("c_concat")
    refs = [r.to(DEV) for r in inp["refs"]]
    ac = lambda: torch.autocast("cuda", dtype=dtype or torch.bfloat16, enabled=dtype is not None)

    def unet(x, t, c):
        with ac():
            return ounet.unet_forward(usd, fc.UNET_CFG, torch.cat([x, cc], 1), t, c, dev("fs")).float()
    x0s = []
    with torch.no_grad():
        fin = osamp.ddim_sample(unet, dev("x_T"), dev("cond"), dev("uncond"), steps, fc.ETA, fc.CFG, fc.RESCALE, sched,
                                noise_fn=lambda i: noises[i], step_callback=lambda i, img, p: x0s.append(p.clone()))
        with ac():
            pix = odec.decode_first_stage(dsd, fin, refs).float()
    return dict(final=fin, x0s=x0s, pix=pix)


def hip_column():
    """per-step HIP error of the last DDIM-50 parity run of this box, if its table is there"""
    for path in (os.path.join(ROOT, "gpurun_out", "ddim50_parity_bf16.txt"),):
        if os.path.exists(path):
            rows = [ln.split() for ln in open(path) if re.match(r"\s*\d+\s+\d", ln)]
            if len(rows) == t50.S:
                return [float(r[2]) for r in rows], path
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--live-fp32", action="store_true", help="re-run the fp32 oracle trajectory instead of using the golden's samples")
    ap.add_argument("--no-ddim", action="store_true", help="single forward + decode only")
    a = ap.parse_args()
    torch.backends.cuda.matmul.allow_tf32 = False
    inp = fc.inputs()
    sd = build_weights()
    usd = sub_state_dict(sd, "model.diffusion_model.")
    dsd = sub_state_dict(sd, "first_stage_model.decoder.")
    dev = lambda k: inp[k].to(DEV)
    print(f"# fp16 vs bf16 autocast floor of the oracle on {torch.cuda.get_device_name(0)}, torch {torch.__version__}")
    print("# (the reference runs under torch.autocast(float16): scripts/evaluation/inference.py:323; this repo's bounds are multiples of the bf16 floor)")

    ts = torch.tensor([fc.UNET_T], device=DEV)
    xin = torch.cat([dev("x_T"), dev("c_concat")], 1)

    def fwd(dtype):
        with torch.no_grad(), torch.autocast("cuda", dtype=dtype or torch.bfloat16, enabled=dtype is not None):
            return ounet.unet_forward(usd, fc.UNET_CFG, xin, ts, dev("cond"), dev("fs")).float()
    y32, y16, ybf = fwd(None), fwd(torch.float16), fwd(torch.bfloat16)
    f16, fbf = rel_l2(y16, y32), rel_l2(ybf, y32)
    print(f"UNet forward (B=1, t={fc.UNET_T}): fp16-autocast floor {f16:.3e} (finite: {bool(torch.isfinite(y16).all())}) | "
          f"bf16-autocast floor {fbf:.3e} | bf16 / fp16 = {fbf / f16:.2f}")

    refs = [r.to(DEV) for r in inp["refs"]]

    def dec(dtype):
        with torch.no_grad(), torch.autocast("cuda", dtype=dtype or torch.bfloat16, enabled=dtype is not None):
            return odec.decode_first_stage(dsd, dev("z_dec"), refs).float()
    d32, d16, dbf = dec(None), dec(torch.float16), dec(torch.bfloat16)
    g16, gbf = rel_l2(d16, d32), rel_l2(dbf, d32)
    print(f"decoder, 16 frames:          fp16-autocast floor {g16:.3e} (finite: {bool(torch.isfinite(d16).all())}) | "
          f"bf16-autocast floor {gbf:.3e} | bf16 / fp16 = {gbf / g16:.2f}")
    del y32, y16, ybf, d32, d16, dbf
    if a.no_ddim:
        return

    g = np.load(t50.GOLDEN)
    run16 = trajectory(usd, dsd, inp, torch.float16, t50.S)
    finite = bool(torch.isfinite(run16["final"]).all() and torch.isfinite(run16["pix"]).all())
    if a.live_fp32:
        run32 = trajectory(usd, dsd, inp, None, t50.S)
        e16 = [rel_l2(run16["x0s"][i], run32["x0s"][i]) for i in range(t50.S)]
        e16_final, e16_pix = rel_l2(run16["final"], run32["final"]), rel_l2(run16["pix"], run32["pix"])
        src = "fp32 oracle re-run in this process, full tensors"
    else:
        ix0, ifin, ipix = t50.sample_positions(run16["x0s"][0].numel(), run16["final"].numel(), run16["pix"].numel())
        pick = lambda t, idx: t.reshape(-1)[idx.to(t.device)].float().cpu()
        e16 = [rel_l2(pick(run16["x0s"][i], ix0), torch.from_numpy(g["x0_fp32"][i])) for i in range(t50.S)]
        e16_final = rel_l2(pick(run16["final"], ifin), torch.from_numpy(g["final_fp32"]))
        e16_pix = rel_l2(pick(run16["pix"], ipix), torch.from_numpy(g["pix_fp32"]))
        src = "fp32 trajectory = tests/golden/ddim50_oracle.npz (sampled positions, as the HIP column of the parity test)"
    fbf = g["floor_x0"].tolist()
    hip, hip_src = hip_column()
    print(f"\n# DDIM-{t50.S} CFG {fc.CFG} eta {fc.ETA} rescale {fc.RESCALE}: rel-L2 of pred_x0 against the fp32 oracle trajectory; {src}")
    print(f"# fp16 trajectory finite: {finite}" + (f"; HIP column: {os.path.relpath(hip_src, ROOT)}" if hip else "; HIP column: not on this box (see profiles/r0N_ddim50_parity_bf16.txt)"))
    print("step  floor(fp16-autocast)  floor(bf16-autocast)  bf16/fp16" + ("  HIP path  HIP/fp16-floor" if hip else ""))
    ratios = []
    for i in range(t50.S):
        r = fbf[i] / e16[i]
        ratios.append(r)
        line = f"{i:4d}  {e16[i]:.3e}  {fbf[i]:.3e}  {r:6.2f}"
        if hip:
            line += f"  {hip[i]:.3e}  {hip[i] / e16[i]:6.2f}"
        print(line)
    print(f"final latent:   fp16 floor {e16_final:.3e} | bf16 floor {float(g['floor_final']):.3e} | bf16 / fp16 = {float(g['floor_final']) / e16_final:.2f}")
    print(f"decoded pixels: fp16 floor {e16_pix:.3e} | bf16 floor {float(g['floor_pix']):.3e} | bf16 / fp16 = {float(g['floor_pix']) / e16_pix:.2f}")
    print(f"median bf16 / fp16 floor ratio over the {t50.S} steps: {sorted(ratios)[t50.S // 2]:.2f} (min {min(ratios):.2f}, max {max(ratios):.2f})")


if __name__ == "__main__":
    main()
