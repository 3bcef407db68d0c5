#!/usr/bin/env python3
"""Which kernel family should each GEMM problem of a B = 2 UNet forward (and of a 16-frame decode) take?  Records every
tc_gemm_bf16 call of one eager forward / decode of the bench model (shapes, gather geometry, epilogue terms, the real
operands), then times each UNIQUE problem under every routing the library offers -- read per call from the
environment -- interleaved, in this one process, and prints where the default heuristic is not the fastest choice.

    python scripts/gemm_autotune.py [--decoder] > gpurun_out/gemm_autotune.txt
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from tooncrafter_amd import ops  # noqa: E402

ENV_KEYS = ("TC_GEMM_TILE", "TC_GEMM_TILE16", "TC_GEMM_WS", "TC_GEMM_PIPE", "TC_GEMM_SPLITK", "TC_GEMM_WIDE", "TC_GEMM8", "TC_CONV_HALO",
            "TC_CONV_HALO_TALL", "TC_CONV_HALO_KSPLIT", "TC_G16_ILV", "TC_G16_TALL")
VARIANTS = [("default", {}),
            ("default, plain K loop", {"TC_GEMM_PIPE": "0"}),
            ("128x128", {"TC_GEMM_TILE": "22", "TC_GEMM_TILE16": "0", "TC_GEMM_WS": "0"}),
            ("64x64", {"TC_GEMM_TILE": "11", "TC_GEMM_TILE16": "0", "TC_GEMM_WS": "0"}),
            ("128x64", {"TC_GEMM_TILE": "21", "TC_GEMM_TILE16": "0", "TC_GEMM_WS": "0"}),
            ("64x128", {"TC_GEMM_TILE": "12", "TC_GEMM_TILE16": "0", "TC_GEMM_WS": "0"}),
            ("256-row", {"TC_GEMM_TILE": "w", "TC_GEMM_TILE16": "0", "TC_GEMM_WS": "0"}),
            ("256-row + pipe", {"TC_GEMM_TILE": "w", "TC_GEMM_TILE16": "0", "TC_GEMM_WS": "0", "TC_GEMM_PIPE": "2"}),
            ("160x160", {"TC_GEMM_TILE16": "2", "TC_GEMM_WS": "0"}),
            ("160x160 + pipe", {"TC_GEMM_TILE16": "2", "TC_GEMM_WS": "0", "TC_GEMM_PIPE": "2"}),
            ("weight-stationary", {"TC_GEMM_WS": "4"}),
            ("no split-K", {"TC_GEMM_SPLITK": "0"}),
            ("split-K 2", {"TC_GEMM_SPLITK": "2"}),
            ("split-K 4", {"TC_GEMM_SPLITK": "4"}),
            ("split-K 8", {"TC_GEMM_SPLITK": "8"}),
            # round 4 / 5 kernels (a tile-family switch also sends the 3x3 convolutions back to the implicit GEMM: conv_halo.hip)
            ("8-wave 256x256", {"TC_GEMM8": "2"}),
            ("no 8-wave", {"TC_GEMM8": "0"}),
            ("160x160, plain loop", {"TC_GEMM_TILE16": "2", "TC_GEMM_WS": "0", "TC_G16_ILV": "0"}),
            ("160x160, loop 2", {"TC_GEMM_TILE16": "2", "TC_GEMM_WS": "0", "TC_G16_ILV": "2"}),
            ("halo patches off", {"TC_CONV_HALO": "0"}),
            ("halo, tall patches", {"TC_CONV_HALO_TALL": "1"}),
            ("halo, K split always", {"TC_CONV_HALO_KSPLIT": "2"}),
            ("halo, no K split", {"TC_CONV_HALO_KSPLIT": "0"})]


def set_env(env):
    for k in ENV_KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)


def main():
    dev = torch.device("cuda:0")
    hip = ops.backend()
    model = bench.build_model(dev)
    inp = bench.make_inputs(dev, 7)
    calls = {}
    real = hip.gemm

    def rec(a, w, bias=None, **kw):
        conv = kw.get("conv")
        key = (tuple(a.shape), tuple(w.shape), None if conv is None else tuple(sorted(conv.items())), kw.get("act", 0),
               kw.get("residual") is not None, kw.get("row_bias") is not None, bool(kw.get("out_f32")), kw.get("batch", 1),
               kw.get("a_norm_eps") is not None, kw.get("m"))
        e = calls.setdefault(key, {"n": 0, "args": (a, w, bias, dict(kw))})
        e["n"] += 1
        res = real(a, w, bias, **kw)
        out = res[0] if isinstance(res, tuple) else res          # gn_stats=True returns (out, partial GroupNorm sums)
        if "out" not in e["args"][3]:
            e["args"][3]["out"] = torch.empty_like(out)          # timed re-runs write here, not into fresh allocations
        return res

    hip.gemm = rec
    with torch.no_grad():
        if "--decoder" in sys.argv:
            z = torch.randn(1, 4, 16, 40, 64, device=dev)
            model.first_stage_model.decoder.use_hipgraph = False
            model.decode_first_stage(z, ref_context=inp["refs"])
        else:
            bench.guided_forward(model, inp)()
    hip.gemm = real
    torch.cuda.synchronize()
    print(hip.lib.tc_build_info().decode(), torch.cuda.get_device_name(0))
    print(f"# {sum(e['n'] for e in calls.values())} tc_gemm_bf16 calls, {len(calls)} unique problems; times in us (median of 3 interleaved rounds x 10)")
    rows, total_def, total_best = [], 0.0, 0.0
    for key, e in calls.items():
        a, w, bias, kw = e["args"]
        times = {name: [] for name, _ in VARIANTS}
        ok = {}
        for name, env in VARIANTS:                      # warm-up + which variants are accepted at all
            set_env(env)
            try:
                real(a, w, bias, **kw)
                torch.cuda.synchronize()
                ok[name] = True
            except Exception:                            # noqa: BLE001  (a forced family that cannot take the problem)
                ok[name] = False
        for _ in range(3):
            for name, env in VARIANTS:
                if not ok[name]:
                    continue
                set_env(env)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    real(a, w, bias, **kw)
                e1.record()
                torch.cuda.synchronize()
                times[name].append(e0.elapsed_time(e1) * 100.0)
        med = {n: sorted(v)[1] for n, v in times.items() if v}
        best = min(med, key=med.get)
        shape = f"{key[0][0]}x{key[1][0]}x{key[1][1]}"
        conv = dict(key[2]) if key[2] else None
        tag = ("lin" if conv is None else conv["kind"]) + (" geglu" if key[3] == 3 else "") + (" +res" if key[4] else "") + \
              (" +rowbias" if key[5] else "") + (" f32" if key[6] else "") + (f" batch{key[7]}" if key[7] != 1 else "") + (" +LN" if key[8] else "")
        rows.append((e["n"] * (med["default"] - med[best]), e["n"], shape, tag, med, best))
        total_def += e["n"] * med["default"]
        total_best += e["n"] * med[best]
    set_env({})
    rows.sort(key=lambda r: -r[0])
    print(f"# sum over calls: default routing {total_def / 1e3:.2f} ms, best-of-all {total_best / 1e3:.2f} ms")
    for gain, n, shape, tag, med, best in rows:
        alts = "  ".join(f"{k} {v:.1f}" for k, v in sorted(med.items(), key=lambda kv: kv[1])[:4])
        print(f"{n:3d} x {shape:22s} {tag:22s} default {med['default']:7.1f} | best: {best:22s} {med[best]:7.1f} (x{med['default'] / med[best]:.3f}, "
              f"{gain / 1e3:6.3f} ms per pass) | {alts}")


if __name__ == "__main__":
    main()
