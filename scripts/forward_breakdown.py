#!/usr/bin/env python3
"""Where one guided (B = 2) UNet forward spends its time, by OPERATOR CALL and SHAPE: every method of the HIP backend that
launches kernels is bracketed with HIP events on the launch stream (eager, un-captured; the host is given a head start so the
gaps are not launch-bound), and calls are grouped by (operator, shape signature).  Complements the rocprofv3 per-kernel table
(profiles/r0N_clip_kernel_stats_*.txt), which knows kernels and grids but not which layer a launch belongs to.

  python scripts/forward_breakdown.py [--decoder] [--top N]     -> one table on stdout
"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from tooncrafter_amd import ops  # noqa: E402

OPS = ["gemm", "ff_geglu_fused", "temporal_attn_fused", "temporal_qkv_attn", "attention", "attention_temporal", "groupnorm", "layernorm",
       "softmax_rows", "nchw_to_rows", "rows_to_nchw", "concat_rows", "repeat_rows", "timestep_embedding", "time_mix3", "gn_conv"]


def sig(name, args, kw, out):
    t = [a for a in args if torch.is_tensor(a)]
    if name == "gemm":
        a, w = args[0], args[1]
        conv = kw.get("conv")
        o = out[0] if isinstance(out, tuple) else out
        m = kw.get("m") or o.shape[0]
        kind = "lin" if conv is None else conv["kind"]
        tags = [kind]
        if kw.get("act"):
            tags.append(f"act{kw['act']}")
        if kw.get("residual") is not None:
            tags.append("+res")
        if kw.get("row_bias") is not None:
            tags.append("+rowbias")
        if kw.get("a_norm_eps") is not None:
            tags.append("+LN")
        if kw.get("batch", 1) != 1:
            tags.append(f"x{kw['batch']}")
        return f"{m}x{w.shape[0]}x{w.shape[1]} {' '.join(tags)}", 2.0 * m * w.shape[0] * w.shape[1] * kw.get("batch", 1)
    if name == "temporal_qkv_attn":
        return f"{args[0].shape[0]}x{args[1].shape[0]}x{args[1].shape[1]} qkv+attn", 2.0 * args[0].shape[0] * args[1].shape[0] * args[1].shape[1]
    if name in ("groupnorm", "gn_conv"):
        return f"s{kw.get('samples')} r{kw.get('rows')} c{t[0].shape[-1]}", 0.0
    if name == "attention":
        return f"b{kw.get('batch')} h{kw.get('heads')} lq{kw.get('lq')} lk{kw.get('lk')}" + (f"+{kw.get('lk2')}" if kw.get("lk2") else ""), \
            4.0 * kw["batch"] * kw["heads"] * kw["lq"] * (kw["lk"] + (kw.get("lk2") or 0)) * 64
    return "x".join(str(d) for d in t[0].shape) if t else "-", 0.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--decoder", action="store_true", help="break down one 16-frame decode instead of the UNet forward")
    ap.add_argument("--top", type=int, default=60)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    os.environ.setdefault("TC_HIPGRAPH", "0")
    model = bench.build_model(dev)
    inp = bench.make_inputs(dev, 7)
    be = ops.backend()
    if args.decoder:
        z = torch.randn(1, 4, 16, 40, 64, device=dev)
        fwd = lambda: model.decode_first_stage(z, ref_context=inp["refs"])          # noqa: E731
    else:
        fwd = bench.guided_forward(model, inp)
    rec = []
    depth = [0]

    def wrap(name):
        orig = getattr(be, name)

        def f(*a, **kw):
            if depth[0]:                                   # an operator called from inside another one (gn_conv -> groupnorm + gemm)
                return orig(*a, **kw)
            depth[0] += 1
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            try:
                out = orig(*a, **kw)
            finally:
                depth[0] -= 1
            e1.record()
            s, fl = sig(name, a, kw, out)
            rec.append((name, s, fl, e0, e1))
            return out
        return orig, f

    with torch.no_grad():
        for _ in range(2):
            fwd()
        torch.cuda.synchronize()
        saved = {}
        for n in OPS:
            if hasattr(be, n) and n != "gn_conv":
                saved[n], f = wrap(n)
                setattr(be, n, f)
        try:
            for _ in range(3):                               # the last pass is the one reported (clocks settled)
                rec.clear()
                torch.cuda._sleep(int(2e7))
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0.record()
                fwd()
                t1.record()
                torch.cuda.synchronize()
        finally:
            for n, o in saved.items():
                setattr(be, n, o)
    tot = collections.OrderedDict()
    for name, s, fl, e0, e1 in rec:
        k = (name, s)
        d = tot.setdefault(k, [0, 0.0, 0.0])
        d[0] += 1
        d[1] += e0.elapsed_time(e1)
        d[2] += fl
    total = sum(d[1] for d in tot.values())
    print(f"# {'decoder 16f' if args.decoder else 'guided UNet forward (B=2)'}: {len(rec)} operator calls, sum of operator times {total:.2f} ms, "
          f"wall (events around the pass) {t0.elapsed_time(t1):.2f} ms")
    byop = collections.Counter()
    for (name, _), d in tot.items():
        byop[name] += d[1]
    print("# by operator: " + "  ".join(f"{n} {ms:.2f}" for n, ms in byop.most_common()))
    print(f"{'operator':22s} {'shape':44s} {'calls':>5s} {'total ms':>9s} {'avg us':>8s} {'TF/s':>7s} {'%':>5s}")
    for (name, s), d in sorted(tot.items(), key=lambda kv: -kv[1][1])[:args.top]:
        tf = f"{d[2] / d[1] / 1e9:7.0f}" if d[2] else "      -"
        print(f"{name:22s} {s:44s} {d[0]:5d} {d[1]:9.3f} {d[1] * 1e3 / d[0]:8.1f} {tf} {100 * d[1] / total:5.1f}")


if __name__ == "__main__":
    main()
