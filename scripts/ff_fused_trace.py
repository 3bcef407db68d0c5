#!/usr/bin/env python3
"""Per-interval timing of csrc/ff_fused.hip (TC_FF_ABLATE=16 build): shader clock after every barrier of block 0's waves 0
(group 0) and 4 (group 1) over three chunks of its second tile.  Prints the interval lengths in cycles."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tooncrafter_amd.lvdm.common import pack_geglu, pack_linear  # noqa: E402
from tooncrafter_amd.ops import HipOps  # noqa: E402

C, HID = 320, 1280
hip = HipOps()
g = torch.Generator().manual_seed(0)
w1, b1 = pack_geglu(torch.randn(2 * HID, C, generator=g) * 0.05, torch.randn(2 * HID, generator=g) * 0.1)
w2, b2 = pack_linear(torch.randn(C, HID, generator=g) * 0.03), torch.randn(C, generator=g) * 0.1
w1, b1, w2, b2 = w1.cuda(), b1.cuda(), w2.cuda(), b2.cuda()
x = (torch.randn(81920, C, generator=g) * 1.5).to(torch.bfloat16).cuda()
for _ in range(3):
    hip.ff_geglu_fused(x, w1, b1, w2, b2, ln_eps=1e-5)
trace = torch.zeros(2, 64, dtype=torch.int64, device="cuda")
os.environ["TC_FF_TRACE"] = hex(trace.data_ptr())
os.environ["TC_FF_ABLATE"] = "16"
hip.ff_geglu_fused(x, w1, b1, w2, b2, ln_eps=1e-5)
torch.cuda.synchronize()
t = trace.cpu()
names = ["R0", "M0", "R1", "M1", "R2", "M2", "R3", "M3", "R4", "M4", "G", "X", "RF", "MF"]
for grp in (0, 1):
    v = t[grp][t[grp] > 0]
    d = (v[1:] - v[:-1]).tolist()
    print(f"group {grp}: {len(v)} stamps; barrier-to-barrier cycles (the label is the interval ENDING at that barrier):")
    # stamp i is taken after barrier i of the traced window; interval i+1 = between stamps i and i+1
    for c in range(3):
        row = []
        for i in range(14):
            k = c * 14 + i - 1
            row.append(f"{names[i]} {d[k]:5d}" if 0 <= k < len(d) else f"{names[i]}     -")
        print("  chunk", c, " | ".join(row), " sum", sum(d[max(0, c * 14 - 1):c * 14 + 13]))
print("clock: s_memtime ticks (100 MHz constant clock on some parts: compare with the kernel's wall time)")
