#!/usr/bin/env python3
"""Calibration of the parity tolerances (SURVEY.md 8d / VERDICT r1 item 3): run the ORACLE code itself on the
GPU twice -- fp32, and under torch.autocast(bfloat16), which is how the reference runs (fp16 autocast,
inference.py:333) -- at the full BASELINE shapes.  The distance between those two is the noise floor of "a
standard reduced-precision PyTorch implementation"; the HIP path's distance to the same fp32 run must stay
within 1.5x of it.  Also times the eager PyTorch-ROCm oracle (the "for free" GPU baseline of SURVEY 8d).

Test infrastructure: imports oracle/, never imported by the product.  Output: one text report.
    python scripts/noise_floor.py > gpurun_out/noise_floor.txt
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import fullsize_cases as fc  # noqa: E402
from conftest import rel_l2, sub_state_dict  # noqa: E402
from oracle import decoder as odec  # noqa: E402
from oracle import sampler as osamp  # noqa: E402
from oracle import unet as ounet  # noqa: E402

DEV = "cuda"


def timed(fn, reps=1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return out, (time.perf_counter() - t0) / reps


def main():
    import bench
    from tooncrafter_amd.utils import instantiate_from_config
    torch.backends.cuda.matmul.allow_tf32 = False
    inp = fc.inputs()
    golden = dict(np.load(fc.GOLDEN_FILE)) if os.path.exists(fc.GOLDEN_FILE) else {}
    dev = lambda k: inp[k].to(DEV)
    print(f"# noise floor calibration on {torch.cuda.get_device_name(0)}, torch {torch.__version__}")

    # ---------------------------------------------------------------- HIP model (CPU-synth weights)
    with torch.device("meta"):
        model = instantiate_from_config(dict(target="lvdm.models.ddpm3d.LatentVisualDiffusion", params=bench.MODEL_PARAMS))
    model = model.to_empty(device=DEV).eval()
    sd_all = {}
    with torch.no_grad():
        for name, p in model.named_parameters():
            v = bench.synth.synth_tensor(name, tuple(p.shape), 1234, "cpu")
            p.copy_(v)
            sd_all[name] = p.detach()                      # fp32 device copy shared with the oracle runs
        bufs = bench.instantiate_schedule()
        for name, b in model.named_buffers():
            b.copy_(bufs[name].to(DEV))
    usd = sub_state_dict(sd_all, "model.diffusion_model.")
    dsd = sub_state_dict(sd_all, "first_stage_model.decoder.")
    un = model.model.diffusion_model
    xin = lambda x: torch.cat([x, dev("c_concat")], 1)
    ts = torch.tensor([fc.UNET_T], device=DEV)

    def oracle_unet(x, t, c, autocast):
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            return ounet.unet_forward(usd, fc.UNET_CFG, xin(x), t, c, dev("fs")).float()

    with torch.no_grad():
        # ------------------------------------------------------------ one UNet forward
        oracle_unet(dev("x_T"), ts, dev("cond"), False)                      # warm (MIOpen / hipBLASLt selection)
        y32, t32 = timed(lambda: oracle_unet(dev("x_T"), ts, dev("cond"), False))
        oracle_unet(dev("x_T"), ts, dev("cond"), True)
        ybf, tbf = timed(lambda: oracle_unet(dev("x_T"), ts, dev("cond"), True))
        hip = lambda: un(None, ts, context=dev("cond"), fs=dev("fs"), x_parts=[dev("x_T"), dev("c_concat")])
        hip()
        yh, th = timed(hip)
        floor = rel_l2(ybf, y32)
        print(f"UNet fwd B=1 t={fc.UNET_T}: oracle bf16-autocast vs oracle fp32 (GPU): rel-L2 {floor:.3e}   <- noise floor")
        print(f"                      HIP path vs oracle fp32 (GPU):            rel-L2 {rel_l2(yh, y32):.3e}  "
              f"({rel_l2(yh, y32) / floor:.2f} x floor)")
        if "unet_y" in golden:
            g = torch.from_numpy(golden["unet_y"])
            print(f"                      oracle fp32 GPU vs committed CPU golden:  rel-L2 {rel_l2(y32.cpu(), g):.3e}")
            print(f"                      HIP path vs committed CPU golden:         rel-L2 {rel_l2(yh.cpu(), g):.3e}")
        print(f"  eager PyTorch-ROCm oracle: fp32 {t32 * 1e3:.1f} ms, bf16 autocast {tbf * 1e3:.1f} ms; HIP path (eager, B=1) {th * 1e3:.1f} ms")

        # ------------------------------------------------------------ 3-step CFG DDIM
        sched = osamp.make_schedule_buffers()

        def oracle_ddim(autocast):
            x0s = []
            fin = osamp.ddim_sample(lambda x, t, c: oracle_unet(x, t, c, autocast), dev("x_T"), dev("cond"),
                                    dev("uncond"), fc.DDIM_STEPS, fc.ETA, fc.CFG, fc.RESCALE, sched,
                                    noise_fn=lambda i: inp["noises"][i].to(DEV),
                                    step_callback=lambda i, img, p: x0s.append(p.clone()))
            return fin, x0s
        try:
            f32, x32 = oracle_ddim(False)
            fbf, xbf = oracle_ddim(True)
            print(f"DDIM-3 CFG 7.5: oracle bf16-autocast vs oracle fp32: pred_x0 per step "
                  f"{[f'{rel_l2(a, b):.3e}' for a, b in zip(xbf, x32)]}, final {rel_l2(fbf, f32):.3e}   <- noise floor")
            if "ddim_final" in golden:
                print(f"                oracle fp32 GPU vs committed CPU golden: final {rel_l2(f32.cpu(), torch.from_numpy(golden['ddim_final'])):.3e}")
        except Exception as e:                                      # device placement inside the CPU oracle
            print(f"DDIM-3 oracle on GPU skipped: {type(e).__name__}: {e}")

        # ------------------------------------------------------------ decoder 16 frames
        refs = [r.to(DEV) for r in inp["refs"]]
        z = dev("z_dec")

        def oracle_dec(autocast):
            st = {}
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                y = odec.decode_first_stage(dsd, z, refs, probe=lambda n, h: st.__setitem__(n, h.float()))
            return y.float(), st
        oracle_dec(False)
        (d32, s32), td32 = timed(lambda: oracle_dec(False))
        oracle_dec(True)
        (dbf, sbf), tdbf = timed(lambda: oracle_dec(True))
        sh = {}
        dec = model.first_stage_model.decoder
        probe = lambda n, act: sh.__setitem__(n, fc.nchw_flat_from_rows(act.rows, act.frames, act.h, act.w).float())
        dec.decode_clip(z, refs, scale=1 / 0.18215)
        dh, tdh = timed(lambda: dec.decode_clip(z, refs, scale=1 / 0.18215, probe=probe))
        print(f"decoder 16f: oracle bf16-autocast vs oracle fp32: out {rel_l2(dbf, d32):.3e}; stages "
              f"{ {n: f'{rel_l2(sbf[n], s32[n]):.3e}' for n in fc.PROBES} }   <- noise floor")
        print(f"             HIP path vs oracle fp32 (GPU):       out {rel_l2(dh, d32):.3e}; stages "
              f"{ {n: f'{rel_l2(sh[n], s32[n].reshape(-1)):.3e}' for n in fc.PROBES} }")
        print(f"  eager PyTorch-ROCm oracle: fp32 {td32 * 1e3:.0f} ms, bf16 autocast {tdbf * 1e3:.0f} ms; HIP path {tdh * 1e3:.0f} ms")
        clip32 = 100 * t32 + 1.9 * td32
        clipbf = 100 * tbf + 1.9 * tdbf
        print(f"eager PyTorch-ROCm baseline, one clip = 100 UNet forwards + 16f + 14f decode (1.9 x 16f): "
              f"fp32 {clip32:.1f} s = {16 / clip32:.3f} frames/s; bf16 autocast {clipbf:.1f} s = {16 / clipbf:.3f} frames/s")


if __name__ == "__main__":
    main()
