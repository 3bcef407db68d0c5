#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc pass behind bench.py's `roofline.traffic`: UNet B=2 forwards of the
benchmark model (eager, so every tc_gemm_bf16 launch is its own dispatch).  argv[1] = number of forwards."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda:0")
model = bench.build_model(dev)
inp = bench.make_inputs(dev, 7)
un = model.model.diffusion_model
x2, cc2 = torch.cat([inp["x_T"]] * 2), torch.cat([inp["c_concat"]] * 2)
ctx2 = torch.cat([inp["cond"], inp["uncond"]])
ts = torch.full((2,), 499, device=dev, dtype=torch.long)
fs2 = torch.cat([inp["fs"]] * 2)
with torch.no_grad():
    fwd = bench.guided_forward(model, inp)        # as apply_model_multi issues it (shared CFG prefix)
    for _ in range(n):
        fwd()
torch.cuda.synchronize()
print("forwards", n)
