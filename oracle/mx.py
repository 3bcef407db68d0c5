"""CPU restatement of the MXFP8 operand format of the fp8 GEMM path (BASELINE.json configs[4]).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  The reference has no fp8 path, so there is no reference
code or golden vector to pin this against: the format is the public OCP Microscaling Formats (MX) v1.0
specification, restated here --

* element type e4m3 (OCP FP8 "e4m3fn": bias 7, max normal 448, no infinities), spec section 5.3.2;
* one shared E8M0 scale X = 2^(byte - 127) per block of k = 32 consecutive elements, section 5.1 / 5.2;
* conversion (section 6.3): shared_exp = floor(log2(max |v|)) - emax_elem with emax_elem = 8 for e4m3,
  X = 2^shared_exp clamped to the E8M0 range, elements P_i = quantize_to_e4m3(v_i / X), round to nearest even,
  values beyond the e4m3 range saturating to +-448 (the spec leaves overflow handling implementation-defined;
  saturation is what tc_quant_mxfp8 does and what this file states);
* dot product of two blocks (section 6.2): X_a X_b sum_i P_a,i P_b,i.

The kernels under test are pinned to THIS restatement bit for bit (the quantiser) and to fp32 accumulation-order
tolerance (the GEMM); the end-to-end error the format introduces against the fp32 oracle of the reference's
algorithm is measured and bounded separately in tests/test_gpu_fp8.py ("parity unpinned" for the format
itself: no reference vectors exist)."""
import torch

BLOCK = 32
E4M3_MAX = 448.0
E4M3_EMAX = 8


def quantize_mxfp8(x: torch.Tensor):
    """x: [rows, k] (k % 32 == 0), any float dtype.  Returns (q uint8 [rows, k] e4m3 bytes,
    s uint8 [rows, k / 32] E8M0 bytes)."""
    rows, k = x.shape
    assert k % BLOCK == 0
    v = x.detach().to(torch.float32).reshape(rows, k // BLOCK, BLOCK)
    amax = v.abs().amax(dim=2)
    _, e = torch.frexp(amax)                          # amax = m * 2^e, m in [0.5, 1): floor(log2 amax) = e - 1
    shared = (e - 1 - E4M3_EMAX).clamp(-127, 127)
    shared = torch.where(amax > 0, shared, torch.full_like(shared, -127))
    # bf16 inputs whose amax is denormal carry exponent field 0: the hardware path sees field 0 -> byte 0
    shared = torch.where(amax < 2.0 ** -126, torch.full_like(shared, -127), shared)
    scaled = torch.ldexp(v, (-shared).unsqueeze(2).expand_as(v).to(torch.int32))
    scaled = scaled.clamp(-E4M3_MAX, E4M3_MAX)
    q = scaled.to(torch.float8_e4m3fn).view(torch.uint8).reshape(rows, k)
    return q, (shared + 127).to(torch.uint8)


def dequantize_mxfp8(q: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
    """(q, s) -> fp32 [rows, k]: P_i * 2^(s - 127), exact."""
    rows, k = q.shape
    p = q.view(torch.float8_e4m3fn).to(torch.float32).reshape(rows, k // BLOCK, BLOCK)
    e = (s[:, :k // BLOCK].to(torch.int32) - 127).unsqueeze(2).expand_as(p)
    return torch.ldexp(p, e).reshape(rows, k)


def fake_quant(x: torch.Tensor) -> torch.Tensor:
    """x rounded through MXFP8 along its last dimension (what the fp8 GEMM sees of an operand)."""
    shape = x.shape
    q, s = quantize_mxfp8(x.reshape(-1, shape[-1]))
    return dequantize_mxfp8(q, s).reshape(shape)


def gemm_mx(a: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """[M, K] x [N, K]^T with both operands rounded through MXFP8, fp64 accumulation (the exact value the
    fp32-accumulating matrix pipe approximates)."""
    return (fake_quant(a).double() @ fake_quant(w).double().t()).float()
