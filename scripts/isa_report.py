#!/usr/bin/env python3
"""Static report from the compiler (no GPU): per kernel of the given csrc/*.hip files -- instructions, VGPRs, scratch, LDS,
and the counts of the instruction classes the round-4 ISA audit looked at (correctly rounded divisions, transcendentals,
MFMAs, LDS / buffer traffic, waits, barriers).  usage: python scripts/isa_report.py norm conv_halo [...]"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "tooncrafter_amd", "csrc")
EXTRA = {"attention": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}
CLASSES = [("mfma", r"v_mfma_"), ("div_scale", r"v_div_scale_f32"), ("exp", r"v_exp_f32"), ("rcp", r"v_rcp_f32"),
           ("ds_read", r"ds_read"), ("ds_write", r"ds_write"), ("buffer/global ld", r"(buffer|global)_load"),
           ("st", r"(buffer|global)_store"), ("waitcnt", r"s_waitcnt"), ("barrier", r"s_barrier")]


def main():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    print(subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout.splitlines()[0])
    for name in sys.argv[1:] or ["norm", "conv_halo"]:
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, name + ".s")
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", f"-I{ROOT}/include", f"-I{CSRC}",
                   *EXTRA.get(name, []), "-S", "--cuda-device-only", os.path.join(CSRC, name + ".hip"), "-o", out]
            subprocess.run(cmd, check=True, capture_output=True)
            asm = open(out).read()
        heads = [(m.start(), m.group(1)) for m in re.finditer(r"^(_Z\w+):", asm, re.M)]
        res = {k: re.findall(r"^; %s: (\d+)" % k, asm, re.M) for k in ("NumVgprs", "ScratchSize", "LDSByteSize")}
        print(f"\n== csrc/{name}.hip: {len(res['NumVgprs'])} kernels")
        print(f"{'kernel':72s} {'instr':>6s} {'VGPR':>5s} {'scr':>4s} {'LDS':>7s}  " + " ".join(f"{c[0]:>9s}" for c in CLASSES))
        k = 0
        for i, (pos, sym) in enumerate(heads):
            end = heads[i + 1][0] if i + 1 < len(heads) else len(asm)
            body = asm[pos:end]
            if "s_endpgm" not in body:
                continue
            body = body[:body.rindex("s_endpgm")]           # the last exit (kernels with an early return have several)
            dem = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
            dem = re.sub(r"\(anonymous namespace\)::", "", dem)
            dem = re.sub(r"\(.*$", "", dem).replace("void ", "")
            n_instr = len(re.findall(r"^\s+[a-z][a-z0-9_]+", body, re.M))
            counts = [len(re.findall(r"^\s+" + pat, body, re.M)) for _, pat in CLASSES]
            print(f"{dem[:72]:72s} {n_instr:6d} {res['NumVgprs'][k]:>5s} {res['ScratchSize'][k]:>4s} {res['LDSByteSize'][k]:>7s}  "
                  + " ".join(f"{c:9d}" for c in counts))
            k += 1


if __name__ == "__main__":
    main()
