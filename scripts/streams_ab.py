#!/usr/bin/env python3
"""usage: streams_ab.py [rounds] [replays]  -- the guided UNet forward of one DDIM step as apply_model_multi issues it
(hipGraph replay), alternating in ONE process on ONE lease between the batch-2 walk behind the shared prefix
(TC_CFG_STREAMS=0) and the two batch-1 walks on their own HIP streams (TC_CFG_STREAMS=1).  Prints ms per guided forward
per arm and round, and how far the two arms' outputs are apart."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
replays = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = "cuda"
model = bench.build_model(dev)
inp = bench.make_inputs(dev, 7)
cond = {"c_crossattn": [inp["cond"]], "c_concat": [inp["c_concat"]]}
uc = {"c_crossattn": [inp["uncond"]], "c_concat": [inp["c_concat"]]}
ts = torch.full((1,), 499, device=dev, dtype=torch.long)
x = inp["x_T"]
res, outs = {0: [], 1: []}, {}
with torch.no_grad():
    for r in range(rounds):
        for mode in (0, 1):
            model.cfg_streams = bool(mode)
            for _ in range(3):                                  # eager call, capturing call, first replay
                o = model.apply_model_multi(x, ts, [cond, uc], fs=inp["fs"])
            torch.cuda.synchronize()
            assert model._cfg_state["graph"] is not None
            outs[mode] = [t.clone() for t in o]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(replays):
                model.apply_model_multi(x, ts, [cond, uc], fs=inp["fs"])
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / replays
            res[mode].append(ms)
            print(f"round {r} TC_CFG_STREAMS={mode}: {ms:8.3f} ms per guided forward", flush=True)
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
print("streams vs batch-2 walk, rel-L2:", [f"{rel(outs[1][k], outs[0][k]):.3e}" for k in range(2)])
a, b = min(res[0]), min(res[1])
print(f"best: batch-2 walk {a:.3f} ms | streams {b:.3f} ms | x{a / b:.3f}")
