"""The TORCH_LIBRARY(tooncrafter) operator layer without a GPU: the library loads next to the kernel library, every
operator is registered with the schema the Python wrapper uses, and the Meta implementations infer the output shapes
and dtypes (what tracing / export would see).  Compute is covered by tests/test_gpu_torch_ops.py."""
import pytest
import torch

from tooncrafter_amd import _lib


@pytest.fixture(scope="module")
def t():
    from tooncrafter_amd import torch_ops
    return torch_ops.load()


def test_operator_namespace_and_abi(t):
    assert int(t.abi_version()) == _lib.TC_ABI_VERSION
    for name in ("gemm", "quant_mxfp8", "gemm_mx", "attention", "attention_temporal", "groupnorm", "layernorm", "ddim_step",
                 "ff_geglu_fused", "temporal_attn_fused", "temporal_qkv_attn", "groupnorm_pf", "layernorm_pf"):
        op = getattr(t, name)
        schema = str(op.default._schema)
        assert schema.startswith(f"tooncrafter::{name}("), schema
    assert "int[] conv" in str(t.gemm.default._schema) and "Tensor? k2" in str(t.attention.default._schema)


def test_meta_kernels_infer_shapes(t):
    bf = dict(dtype=torch.bfloat16, device="meta")
    f32 = dict(dtype=torch.float32, device="meta")
    a, w = torch.empty(81920, 320, **bf), torch.empty(960, 320, **bf)
    assert t.gemm(a, w, None, None, None, 0, 0, 1.0, 1.0, False, []).shape == (81920, 960)
    wg = torch.empty(2560, 320, **bf)                                            # GEGLU halves the columns
    y = t.gemm(a, wg, torch.empty(2560, **f32), None, None, 0, 3, 1.0, 1.0, False, [])
    assert y.shape == (81920, 1280) and y.dtype == torch.bfloat16
    wc = torch.empty(320, 2880, **bf)                                            # 3x3 conv, stride 2: M from the geometry
    conv = [1, 320, 32, 1, 40, 64, 20, 32, 2, 0, 1]
    y = t.gemm(a, wc, None, None, None, 0, 0, 1.0, 1.0, True, conv)
    assert y.shape == (32 * 20 * 32, 320) and y.dtype == torch.float32
    aq, asc = t.quant_mxfp8(a, 320)                                              # MXFP8 pair: bytes + one scale per 32 K
    assert aq.shape == (81920, 320) and asc.shape == (81920, 12) and aq.dtype == asc.dtype == torch.uint8
    wq, wsc = t.quant_mxfp8(w, 320)
    assert t.gemm_mx(aq, asc, wq, wsc, None, None, None, 0, 0, 1.0, 1.0, False, []).shape == (81920, 960)
    q, kv = torch.empty(32 * 2560, 320, **bf), torch.empty(2 * 77, 320, **bf)
    ki = torch.empty(32 * 16, 320, **bf)
    assert t.attention(q, kv, kv, 32, 5, 2560, 77, 16, 0.125, ki, ki, 16, 1).shape == (32 * 2560, 320)
    assert t.attention_temporal(torch.empty(2 * 16 * 40, 3 * 128, **bf), 2, 16, 40, 2, 0.125).shape == (2 * 16 * 40, 128)
    x = torch.empty(5120, 1280, **bf)
    g = torch.empty(1280, **f32)
    assert t.groupnorm(x, g, g, 32, 160, 1e-5, True).shape == x.shape and t.layernorm(x, g, g, 1e-5).dtype == torch.bfloat16
    wpf = [torch.empty(1280, 11520, **bf), torch.empty(1280, 1280, **bf)]       # ABI 12: a consumer's weights ride on the norm
    assert t.groupnorm_pf(x, g, g, 32, 160, 1e-5, True, wpf).shape == x.shape and t.layernorm_pf(x, g, g, 1e-5, wpf).shape == x.shape
    assert "Tensor[] prefetch" in str(t.groupnorm_pf.default._schema)
    # the level-0 one-launch operators (ABI 9): rows in, rows out
    x0r = torch.empty(81920, 320, **bf)
    y = t.ff_geglu_fused(x0r, torch.empty(2560, 320, **bf), torch.empty(2560, **f32), torch.empty(320, 1280, **bf), torch.empty(320, **f32), 1e-5)
    assert y.shape == (81920, 320) and y.dtype == torch.bfloat16
    y = t.temporal_attn_fused(x0r, torch.empty(960, 320, **bf), torch.empty(960, **f32), torch.empty(320, 320, **bf), torch.empty(320, **f32),
                              2, 16, 2560, 5, 1e-5, 0.125)
    assert y.shape == (81920, 320)
    # ABI 13: qkv projection + temporal attention as one launch (levels 1-3): rows in, rows out, the bias optional
    x1r = torch.empty(20480, 640, **bf)
    y = t.temporal_qkv_attn(x1r, torch.empty(1920, 640, **bf), None, 2, 16, 640, 10, 0.125)
    assert y.shape == (20480, 640) and y.dtype == torch.bfloat16
    assert t.temporal_qkv_attn(x1r, torch.empty(1920, 640, **bf), torch.empty(1920, **f32), 2, 16, 640, 10, 0.125).shape == (20480, 640)
    assert "Tensor? bqkv" in str(t.temporal_qkv_attn.default._schema)
    lat = torch.empty(1, 4, 16, 40, 64, **f32)
    xp, x0 = t.ddim_step(lat, lat, lat, lat, None, 7.5, 7.5, 0.7, 0.6, 0.8, 0.7, 0.5, 0.3, 0.98)
    assert xp.shape == lat.shape and x0.shape == lat.shape


def test_cpu_tensors_are_refused(t):
    a, w = torch.zeros(8, 64, dtype=torch.bfloat16), torch.zeros(64, 64, dtype=torch.bfloat16)
    with pytest.raises((RuntimeError, NotImplementedError)):                     # no CPU kernel is registered: no fallback
        t.gemm(a, w, None, None, None, 0, 0, 1.0, 1.0, False, [])
