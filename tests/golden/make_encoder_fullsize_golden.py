#!/usr/bin/env python3
"""Full-size golden for the first-stage ENCODER, made from the REAL reference and cross-checked
against the repo's oracle in the same run (build container only: needs /root/reference; no GPU).

The reference's `AutoencoderKL_Dualref` (lvdm/models/autoencoder.py:100-110 + lvdm/modules/networks/ae_modules.py:366-475)
is instantiated from the first_stage_config of the UNMODIFIED configs/inference_512_v1.0.yaml (ch = 128), filled with the
synthetic weights every other full-size case uses (synth seed 1234), and run in fp32 on the CPU on 16 frames of
3 x 320 x 512 -- the call scripts/evaluation/inference.py:164-178 makes (`get_latent_z_with_hidden_states`).  Recorded at
fixed pseudo-random positions: the posterior mean and log-variance (before the 0.18215 scale) and the five hidden states
the decoder's reference attention consumes.  oracle/encoder.py runs beside it; the table of their distances goes to
profiles/r04_encoder_fullsize_golden_vs_reference.txt.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_encoder_fullsize_golden.py     # ~2 min, -> encoder_fullsize.npz
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
import make_golden as mg  # noqa: E402

REF = mg.REF
OUT = os.path.join(HERE, "encoder_fullsize.npz")


def main():
    mg.install_stubs()
    sys.path.insert(0, REF)
    sys.path.insert(1, REPO)
    sys.path.insert(2, os.path.join(REPO, "tests"))
    import fullsize_cases as fc
    from conftest import rel_l2, sub_state_dict
    from oracle import encoder as oenc
    from utils.utils import instantiate_from_config            # reference
    torch.set_grad_enabled(False)
    torch.set_num_threads(int(os.environ.get("TC_THREADS", os.cpu_count() or 1)))
    t0 = time.time()
    log = lambda s: print(f"[{time.time() - t0:6.0f}s] {s}", flush=True)

    with open(os.path.join(REF, "configs", "inference_512_v1.0.yaml")) as f:
        cfg = mg.wrap(yaml.safe_load(f))
    ae = instantiate_from_config(cfg.model.params.first_stage_config).eval()
    import lvdm.models.autoencoder as ref_ae
    assert ref_ae.__file__.startswith(REF), ref_ae.__file__
    sd = sub_state_dict(fc.full_state_dict(("first_stage_model.",)), "first_stage_model.")
    missing, unexpected = ae.load_state_dict(sd, strict=False)
    assert not unexpected and not [k for k in missing if k.startswith(("encoder.", "quant_conv."))], (missing[:4], unexpected[:4])
    x = fc.encoder_frames()
    log("reference AutoencoderKL_Dualref from the unmodified YAML; encoding 16 x 3 x 320 x 512")
    post, hidden = ae.encode(x, return_hidden_states=True)
    log("reference done; oracle")
    _, o_mean, o_logvar, o_hidden = oenc.encode(sd, x)
    log("oracle done")

    out = dict(meta=np.array(f"reference encoder, fp32 CPU, torch {torch.__version__}; frames = fullsize_cases.encoder_frames() seed 4242"))
    rows = []

    def put(name, ref_t, orc_t, seed):
        flat = ref_t.reshape(-1)
        idx = fc.sample_idx(flat.numel(), fc.N_ENC, seed)
        out[name] = flat[idx].float().numpy()
        out[name + "_norm"] = np.float64(float(ref_t.double().norm()))
        out[name + "_shape"] = np.array(ref_t.shape)
        e = rel_l2(orc_t, ref_t)
        rows.append(f"{name:10s} shape {tuple(ref_t.shape)}  oracle vs reference rel-L2 {e:.3e}")
        log(rows[-1])
        return e

    worst = put("mean", post.mean, o_mean, 31)
    worst = max(worst, put("logvar", post.logvar, o_logvar, 32))
    assert len(hidden) == len(o_hidden) == 5
    for i, (h, oh) in enumerate(zip(hidden, o_hidden)):
        worst = max(worst, put(f"hid{i}", h, oh, 40 + i))
    np.savez(OUT, **out)
    text = "\n".join(["# first-stage encoder, 16 x 3 x 320 x 512, fp32 CPU: the REAL reference (unmodified inference_512_v1.0.yaml "
                      "first_stage_config) vs oracle/encoder.py, whole tensors; the golden file holds the reference's values",
                      *rows, f"worst {worst:.3e}; wall {time.time() - t0:.0f} s"])
    with open(os.path.join(REPO, "profiles", "r04_encoder_fullsize_golden_vs_reference.txt"), "w") as f:
        f.write(text + "\n")
    print(text)
    print(f"wrote {OUT}: {os.path.getsize(OUT) / 1e6:.2f} MB")
    assert worst < 1e-4, worst


if __name__ == "__main__":
    main()
