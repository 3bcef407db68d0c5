#!/usr/bin/env python3
"""Full-size goldens from the fp32 CPU ORACLE (oracle/, itself pinned to the real reference at tiny
size by make_golden.py / test_oracle_golden.py).  Run here, on the CPU, once (20-40 min on 8 cores):

    python tests/golden/make_fullsize_golden.py            # -> tests/golden/fullsize_oracle.npz

Cases (tests/fullsize_cases.py holds the seeds and the sampling positions):
  unet_y          one UNet forward, B=1, t=601, cond context                     (whole tensor)
  ddim_pred_x0_i  3-step DDIM, CFG 7.5, rescale 0.7, eta 1, uniform_trailing, injected noise:
  ddim_final      pred_x0 after every step and the final latent                  (whole tensors)
  dec16_out/_<stage>  VideoDecoder, 16 frames at 40x64 latents -> 320x512, the output and the activation
  dec14_out/_<stage>  after the mid block and after every level's reference fusion; same for the 14-frame
                      re-decode (inference.py:264-267)                            (sampled positions)
Nothing here reads /root/reference; the GPU box replays the file without the oracle.
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import fullsize_cases as fc  # noqa: E402
from conftest import sub_state_dict  # noqa: E402
from oracle import decoder as odec  # noqa: E402
from oracle import sampler as osamp  # noqa: E402
from oracle import unet as ounet  # noqa: E402


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    out = {}
    inp = fc.inputs()
    t0 = time.time()
    only = set(sys.argv[1:])
    prev = dict(np.load(fc.GOLDEN_FILE)) if os.path.exists(fc.GOLDEN_FILE) else {}

    if not only or "unet" in only or "ddim" in only:
        usd = sub_state_dict(fc.full_state_dict(("model.diffusion_model.",)), "model.diffusion_model.")
        print(f"[{time.time() - t0:6.0f}s] UNet weights ready", flush=True)
        xin = lambda x: torch.cat([x, inp["c_concat"]], 1)
        with torch.no_grad():
            if not only or "unet" in only:
                y = ounet.unet_forward(usd, fc.UNET_CFG, xin(inp["x_T"]), torch.tensor([fc.UNET_T]), inp["cond"], inp["fs"])
                out["unet_y"] = y.numpy()
                print(f"[{time.time() - t0:6.0f}s] unet_y std {float(y.std()):.4f}", flush=True)
            if not only or "ddim" in only:
                bufs = osamp.make_schedule_buffers()
                x0s = []
                apply = lambda x, ts, c: ounet.unet_forward(usd, fc.UNET_CFG, xin(x), ts, c, inp["fs"])

                def cb(i, img, pred_x0):
                    x0s.append(pred_x0.clone())
                    print(f"[{time.time() - t0:6.0f}s] ddim step {i} pred_x0 std {float(pred_x0.std()):.4f}", flush=True)
                final = osamp.ddim_sample(apply, inp["x_T"], inp["cond"], inp["uncond"], fc.DDIM_STEPS, fc.ETA, fc.CFG,
                                          fc.RESCALE, bufs, noise_fn=lambda i: inp["noises"][i], step_callback=cb)
                out["ddim_final"] = final.numpy()
                for i, p in enumerate(x0s):
                    out[f"ddim_pred_x0_{i}"] = p.numpy()
        del usd

    if not only or "dec" in only:
        dsd = sub_state_dict(fc.full_state_dict(("first_stage_model.decoder.",)), "first_stage_model.decoder.")
        for tag, z in (("dec16", inp["z_dec"]), ("dec14", inp["z_dec"][:, :, fc.IDX14].contiguous())):
            stages = {}
            with torch.no_grad():
                y = odec.decode_first_stage(dsd, z, inp["refs"], probe=lambda n, h: stages.__setitem__(n, h))
            flat = y.reshape(-1)
            out[f"{tag}_out"] = flat[fc.sample_idx(flat.numel(), fc.N_OUT, 1)].numpy()
            out[f"{tag}_out_norm"] = np.float64(float(y.double().norm()))
            for n, h in stages.items():
                hf = h.reshape(-1)
                out[f"{tag}_{n}"] = hf[fc.sample_idx(hf.numel(), fc.N_PROBE, 2)].numpy()
            print(f"[{time.time() - t0:6.0f}s] {tag}: out std {float(y.std()):.4f}, stages {list(stages)}", flush=True)

    prev.update(out)
    np.savez(fc.GOLDEN_FILE, **prev)
    print(f"wrote {fc.GOLDEN_FILE}: {os.path.getsize(fc.GOLDEN_FILE) / 1e6:.2f} MB, keys {sorted(prev)}")


if __name__ == "__main__":
    main()
