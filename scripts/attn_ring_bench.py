#!/usr/bin/env python3
"""A/B of the K/V staging of attn_d64_dma_kernel: TC_ATTN_RING=2 (two stages, one tile in flight, vmcnt(0) +
__syncthreads per tile) vs 3 (three-stage ring, two tiles in flight, counted vmcnt + raw barrier).  Interleaved rounds.

    python scripts/attn_ring_bench.py > gpurun_out/attn_ring_bench.txt
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tooncrafter_amd import ops  # noqa: E402

DEV, BF = "cuda", torch.bfloat16
hip = ops.backend()


def case(tag, batch, heads, lq, lk, kv_bdiv=1, lk2=0):
    hd = heads * 64
    kvb = (batch + kv_bdiv - 1) // kv_bdiv
    q = torch.randn(batch * lq, hd, device=DEV).to(BF)
    k, v = (torch.randn(kvb * lk, hd, device=DEV).to(BF) for _ in range(2))
    kw = {}
    if lk2:
        kw = dict(k2=torch.randn(batch * lk2, hd, device=DEV).to(BF), v2=torch.randn(batch * lk2, hd, device=DEV).to(BF),
                  lk2=lk2, kv2_bdiv=1)
    out = torch.empty_like(q)

    def run(mode):
        os.environ["TC_ATTN_RING"] = mode
        hip.attention(q, k, v, batch=batch, heads=heads, lq=lq, lk=lk, kv_bdiv=kv_bdiv, out=out, **kw)

    res = {"2": [], "3": []}
    outs = {}
    for mode in res:
        run(mode)
        outs[mode] = out.clone()
    same = torch.equal(outs["2"], outs["3"])
    torch.cuda.synchronize()
    for _ in range(5):
        for mode in res:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(12):
                run(mode)
            e1.record()
            torch.cuda.synchronize()
            res[mode].append(e0.elapsed_time(e1) / 12 * 1e3)
    t2, t3 = (sorted(res[m])[2] for m in ("2", "3"))
    fl = 4.0 * batch * heads * lq * (lk + lk2) * 64
    print(f"{tag:22s} b={batch} h={heads} lq={lq} lk={lk}{'+' + str(lk2) if lk2 else ''}: two stages {t2:8.1f} us {fl / t2 / 1e6:7.1f} TF/s | "
          f"ring of three {t3:8.1f} us {fl / t3 / 1e6:7.1f} TF/s | x{t2 / t3:.3f} | bit-identical {same}", flush=True)


if __name__ == "__main__":
    print(hip.lib.tc_build_info().decode(), torch.cuda.get_device_name(0))
    case("L0 self", 32, 5, 2560, 2560)
    case("L1 self", 32, 10, 640, 640)
    case("L2 self", 32, 20, 160, 160)
    case("L0 text + image", 32, 5, 2560, 77, kv_bdiv=16, lk2=256)
    case("L1 text + image", 32, 10, 640, 77, kv_bdiv=16, lk2=256)
    case("decoder ref fusion", 16, 8, 10240, 20480, kv_bdiv=16)
    case("decoder ref fusion 14f", 14, 8, 10240, 20480, kv_bdiv=14)
