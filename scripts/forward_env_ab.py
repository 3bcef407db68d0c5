#!/usr/bin/env python3
"""Same-process A/B of ONE switch on the guided (B = 2) UNet forward of the bench model: two hipGraphs of the SAME forward,
captured with VAR=A and VAR=B (the host and the library read these switches per call, so the captured launches differ),
replayed alternately in sustained runs, several rounds; every round's sign is printed.  Also: how far the two forwards'
outputs are apart.

usage: python scripts/forward_env_ab.py VAR A B [--runs 12] [--rounds 5]     e.g.  TC_QKV_ATTN 0 1
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from tooncrafter_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("var")
    ap.add_argument("a")
    ap.add_argument("b")
    ap.add_argument("--runs", type=int, default=12)
    ap.add_argument("--rounds", type=int, default=5)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    with torch.no_grad():
        model = bench.build_model(dev)
        inp = bench.make_inputs(dev, 7)
        fwd = bench.guided_forward(model, inp)
        be = ops.backend()
        print(f"# {args.var}: A = {args.a}, B = {args.b}; binding {getattr(be, 'binding', 'ctypes')}; "
              f"{be.lib.tc_build_info().decode()}")
        graphs, outs = {}, {}
        for v in (args.a, args.b):
            os.environ[args.var] = v
            o = fwd()
            torch.cuda.synchronize()
            outs[v] = [t.clone() for t in (o if isinstance(o, (tuple, list)) else [o])]
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                fwd()
            graphs[v] = g
        os.environ.pop(args.var, None)
        rel = [float((x.float() - y.float()).norm() / (y.float().norm() + 1e-30)) for x, y in zip(outs[args.a], outs[args.b])]
        print(f"# outputs A vs B: rel-L2 {['%.3e' % r for r in rel]} (bit-identical: {all(torch.equal(x, y) for x, y in zip(outs[args.a], outs[args.b]))})")

        def run(g):
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.runs):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / args.runs
        for g in graphs.values():                      # settle the clocks on this kernel mix
            for _ in range(3):
                run(g)
        ta, tb = [], []
        for rd in range(args.rounds):
            a = run(graphs[args.a])
            b = run(graphs[args.b])
            ta.append(a)
            tb.append(b)
            print(f"round {rd}: A {a:7.3f} ms | B {b:7.3f} ms | B vs A {100.0 * (a / b - 1.0):+5.2f} %")
        ma, mb = sorted(ta)[len(ta) // 2], sorted(tb)[len(tb) // 2]
        print(f"median: A {ma:.3f} ms | B {mb:.3f} ms | B vs A {100.0 * (ma / mb - 1.0):+.2f} % per guided forward")


if __name__ == "__main__":
    main()
