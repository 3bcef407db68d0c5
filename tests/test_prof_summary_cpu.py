"""scripts/prof_summary.py on a synthetic rocpd database: the tables hold only the product's kernels inside the product window, the
calibration matmuls and the model-build kernels are reported and left out, and the GEMM-family fraction is the
one the trace's durations imply -- the tool that makes `roofline.frac` recomputable from a tracked file is itself checked."""
import json
import os
import sqlite3
import subprocess
import sys

from conftest import ROOT


def _db(path):
    c = sqlite3.connect(path)
    c.execute("create table kernels (name text, start integer, end integer, duration integer, grid_x integer, grid_y integer, grid_z integer)")
    rows = []
    t = 0

    def add(name, us, grid=(256, 1, 1)):
        nonlocal t
        rows.append((name, t, t + us * 1000, us * 1000, *grid))
        t += us * 1000 + 100
    for _ in range(5):
        add("void at::native::(anonymous namespace)::distribution_elementwise_grid_stride_kernel<float>(...)", 7)   # model build: before the window (an ATen kernel in an anonymous namespace of its own)
    add("Cijk_Ailk_Bljk_BBS_BH_MT256x256x64", 900)                                                   # calibration, before
    for step in range(4):                                                                            # 4 "DDIM steps"
        add("void (anonymous namespace)::gemm_kernel<0, 2, 2, true>(TcGemmParams, int, int, int)", 600)
        add("void (anonymous namespace)::conv_halo_kernel<1, 2, 1>(TcGemmParams, int)", 300)
        add("(anonymous namespace)::qkv_attn_kernel((anonymous namespace)::QaArgs)", 100)
        add("_ZN12_GLOBAL__N_115gn_apply_kernelILb1EEEvPKDF16b", 200)
        add("void at::native::(anonymous namespace)::distribution_elementwise_grid_stride_kernel<float>(...)", 5)   # the sampler's randn: inside
        add("(anonymous namespace)::ddim_apply_kernel(TcDdimParams, double const*)", 10)
    add("Cijk_Ailk_Bljk_BBS_BH_MT256x256x64", 900)                                                   # calibration, inside the window
    add("void (anonymous namespace)::gemm_wide_kernel<1, 4, false>(TcGemmParams, int)", 1000)          # a decode
    add("(anonymous namespace)::video_to_u8_kernel(float const*, unsigned char*, int, long)", 5)      # the clip's last kernel
    add("void (anonymous namespace)::gemm_kernel<1, 2, 2, true>(TcGemmParams, int, int, int)", 400)    # the encoder pass bench.py times afterwards
    add("void at::native::vectorized_elementwise_kernel<4, at::native::FillFunctor<float>>(...)", 3)  # an extra, after the window
    c.executemany("insert into kernels values (?,?,?,?,?,?,?)", rows)
    c.commit()
    c.close()


def test_tables_and_fraction(tmp_path):
    db = str(tmp_path / "trace.db")
    _db(db)
    log = tmp_path / "bench.log"
    log.write_text("noise\n" + json.dumps({"value": 25.0, "unit": "frames/s", "lease_calibration": {
        "before_timed_region": {"matmul_8192_bf16_tflops": 1190.0, "copy_1gib_gbs": 5100.0},
        "after_timed_region": {"matmul_8192_bf16_tflops": 1210.0, "copy_1gib_gbs": 5200.0}}}) + "\n")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "prof_summary.py"), db, "20", str(log), "4"],
                         capture_output=True, text=True, check=True).stdout
    assert "[1190.0, 1210.0] TF/s" in out and "ddim_steps 4" in out
    assert "2 calibration matmuls (Cijk_*)" in out and "6 kernels outside the window" in out
    table = out.split("# GEMM family")[0].split("avg_us")[1]          # the per-kernel table proper (below its header line)
    assert "Cijk_" not in table and "FillFunctor" not in table
    assert "distribution_elementwise" in table                      # the sampler's own randn inside the window stays
    # GEMM family: 4 x (600 + 300 + 100) us + 1000 us = 5.0 ms for 4 x 22.661 + 57.32 TFLOP
    fam_ms = 5.4
    tf = (4 * 22.661 + 57.32) / (fam_ms / 1e3)
    line = [ln for ln in out.splitlines() if ln.startswith("# GEMM family")][0]
    assert "14 launches" in line and f"{fam_ms:.1f} ms" in line and f"{tf:.0f} TF/s" in line and f"{tf / 2500.0:.3f} of the" in line
    # per step: 2 usable intervals between ddim_apply marks (the first is skipped), 1.0 ms of GEMM family each
    step = [ln for ln in out.splitlines() if ln.startswith("# per DDIM step")][0]
    assert "(2 steps" in step and "GEMM family 1.00 ms" in step
    # the clip as bench.py composes it: 50 x the 1.00-ms step + the decode behind the last step (1 launch, 1.0 ms)
    clip = [ln for ln in out.splitlines() if ln.startswith("# GEMM family, one clip")][0]
    tfc = (50 * 22.661 + 57.32) / (51.0 / 1e3)
    assert "in 51.0 ms" in clip and f"{tfc:.0f} TF/s" in clip and f"{tfc / 2500.0:.3f} of peak" in clip
