#!/usr/bin/env python3
"""A/B of the ABI-12 weight prefetch (tc_groupnorm_pf / tc_layernorm_pf: the norm in front of a GEMM streams that GEMM's
weights into the Infinity Cache) on the guided B = 2 UNet forward of the bench model: two hipGraphs of the SAME forward,
one captured with the prefetch lists, one without, replayed alternately (sustained runs of 12, 4 rounds), timed by the
in-stream 100 MHz counter (scripts/probes/clock_probe.hip).  Also: the two forwards' outputs must be bit-identical, and
how many norm launches carry a list / how many bytes per forward.
usage: python scripts/prefetch_ab.py > gpurun_out/TAG/prefetch_ab.txt"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from tooncrafter_amd import ops

dev = torch.device("cuda:0")
lib = ctypes.CDLL(os.path.join(ROOT, "scripts", "bin", "libclock_probe.so"))
lib.clk_probe.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
lib.clk_probe.restype = ctypes.c_int
slots = torch.zeros(1024, 2, dtype=torch.int64, device=dev)
_next = [0]


def probe():
    i = _next[0]
    _next[0] += 1
    assert lib.clk_probe(slots[i].data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    return i


with torch.no_grad():
    model = bench.build_model(dev)
    inp = bench.make_inputs(dev, 7)
    fwd = bench.guided_forward(model, inp)
    hip = ops.backend()
    print(f"# binding: {getattr(hip, 'binding', 'ctypes')}; prefetch rule: rows <= {hip.prefetch_max_rows}, tensors >= {hip.prefetch_min_bytes >> 20} MB")
    # census of one eager forward
    seen = []
    orig = hip.prefetch_list

    def counting(x_rows, tensors):
        out = orig(x_rows, tensors)
        seen.append((x_rows, sum(t.numel() * t.element_size() for t in out), len(out)))
        return out
    hip.prefetch_list = counting
    hip.prefetch_on = True
    y_on = fwd()
    hip.prefetch_list = orig
    with_list = [s for s in seen if s[2]]
    print(f"# norm launches offered a list: {len(seen)}; carrying one: {len(with_list)}; bytes prefetched per forward: "
          f"{sum(s[1] for s in with_list) / 2 ** 20:.0f} MB in {sum(s[2] for s in with_list)} tensors")
    hip.prefetch_on = False
    y_off = fwd()
    torch.cuda.synchronize()
    same = all(torch.equal(a, b) for a, b in zip((y_on if isinstance(y_on, (tuple, list)) else [y_on]),
                                                 (y_off if isinstance(y_off, (tuple, list)) else [y_off])))
    print(f"# outputs of the two forwards bit-identical: {same}")
    # variants of the rule: (label, on, max rows of the consumer, smallest tensor worth a request, plan)
    variants = [("off", False, 8192, 1 << 20, 2), ("on", True, 8192, 1 << 20, 2)]
    if len(sys.argv) > 1 and sys.argv[1] == "sweep":
        variants += [("on, rows <= 20480 (level 1 too)", True, 20480, 1 << 20, 2), ("on, every level", True, 1 << 30, 1 << 20, 2),
                     ("on, tensors >= 256 KB", True, 8192, 256 << 10, 2)]
    if len(sys.argv) > 1 and sys.argv[1] == "plans":     # with scripts/experiments/prefetch_plan2.patch.txt applied (its plan 2 = the default there)
        variants += [("on, plan 1 (every norm its own consumer)", True, 8192, 1 << 20, 1)]
    graphs, outs = {}, {}
    for name, flag, max_rows, min_bytes, plan in variants:
        hip.prefetch_on, hip.prefetch_max_rows, hip.prefetch_min_bytes, hip.prefetch_plan = flag, max_rows, min_bytes, plan
        outs[name] = fwd()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fwd()
        graphs[name] = g
    print("# outputs bit-identical across the variants:", all(torch.equal(outs["off"], o) for o in outs.values()))
    hip.prefetch_on, hip.prefetch_max_rows, hip.prefetch_min_bytes, hip.prefetch_plan = True, 8192, 1 << 20, 2
    N = 12
    rec = {k: [] for k in graphs}
    order = list(graphs)
    for rnd in range(5):
        for k in (order if rnd % 2 == 0 else order[::-1]):
            g = graphs[k]
            for _ in range(4):
                g.replay()
            a = probe()
            for _ in range(N):
                g.replay()
            rec[k].append((a, probe()))
    torch.cuda.synchronize()
    s = slots.cpu()
    ms = {k: [float(s[b, 1] - s[a, 1]) / 100e3 / N for a, b in v] for k, v in rec.items()}
    mean = {k: sum(v[1:]) / len(v[1:]) for k, v in ms.items()}
    for k, v in ms.items():
        print(f"prefetch {k:34s}: " + "  ".join(f"{x:7.3f}" for x in v) + f"   ms per forward; rounds 2-5 mean {mean[k]:7.3f}  ({100 * (mean['off'] / mean[k] - 1):+.2f} %)")
    print(f"forward: {mean['off']:.3f} -> {mean['on']:.3f} ms ({100 * (mean['off'] / mean['on'] - 1):+.2f} % throughput)")
