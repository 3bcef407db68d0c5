#!/usr/bin/env python3
"""Per-interval timing of csrc/tb_fused.hip (TC_TB_ABLATE=8 + TC_TB_TRACE): shader clock after every barrier of block 0's waves 0
(group 0) and 4 (group 1) over the first two heads of its second tile.  Prints the interval lengths in cycles."""
import os
import sys

import torch

os.environ.setdefault("TC_TB_FUSED", "1")   # opt-in since round 6 (the level-0 default is the chain around csrc/qkv_attn.hip)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tooncrafter_amd.lvdm.common import pack_linear  # noqa: E402
from tooncrafter_amd.ops import HipOps  # noqa: E402

C, HEADS, T = 320, 5, 16
hip = HipOps()
g = torch.Generator().manual_seed(0)
b, hw = 2, 2560
wqkv = pack_linear(torch.randn(3 * C, C, generator=g) * 0.06).cuda()
bqkv = (torch.randn(3 * C, generator=g) * 0.1).cuda()
wo, bo = pack_linear(torch.randn(C, C, generator=g) * 0.05).cuda(), (torch.randn(C, generator=g) * 0.1).cuda()
x = (torch.randn(b * T * hw, C, generator=g) * 1.5).to(torch.bfloat16).cuda()
run = lambda: hip.temporal_attn_fused(x, wqkv, bqkv, wo, bo, b=b, t=T, hw=hw, heads=HEADS, ln_eps=1e-5)
for _ in range(3):
    run()
trace = torch.zeros(2, 64, dtype=torch.int64, device="cuda")
os.environ["TC_TB_TRACE"] = hex(trace.data_ptr())
os.environ["TC_TB_ABLATE"] = "8"
run()
torch.cuda.synchronize()
t = trace.cpu()
names = [f"{s}{k}" for k in range(5) for s in ("Ra", "Ma")] + ["GA"] + [f"{s}{k}" for k in range(5) for s in ("Rb", "Mb")] + ["GB", "ATT", "RF", "MF"]
for grp in (0, 1):
    v = t[grp][t[grp] > 0]
    d = (v[1:] - v[:-1]).tolist()
    print(f"group {grp}: {len(v)} stamps; cycles between barriers, labelled by the interval that ENDS at the barrier (25 per head)")
    for h in range(2):
        row = []
        for i in range(25):
            k = h * 25 + i - 1
            row.append(f"{names[i]} {d[k]}" if 0 <= k < len(d) else f"{names[i]} -")
        print(f"  head {h}:", " | ".join(row), " sum", sum(d[max(0, h * 25 - 1):h * 25 + 24]))
