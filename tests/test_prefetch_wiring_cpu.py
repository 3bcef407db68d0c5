"""ABI 12 host wiring, on the CPU: which weights the module mirror hands to its norm launches as prefetch lists.

The prefetch is a cache hint -- it cannot change a result, so no parity test would notice a wrong list.  What a wrong list
costs is time (weights streamed that nobody reads soon, or the ones that matter left cold), so the invariant is pinned here
with the emulated operator contract (tests/emu_ops.py) recording every call of a tiny UNet forward:
  * every tensor a norm is asked to stream IS (object identity) the weight operand of a GEMM among the next GEMM calls;
  * a GroupNorm in front of a convolution names that convolution's weights first, a transformer block's LayerNorm the
    projection behind it and the one after that;
  * the rule that decides what is actually streamed (ops.HipOps.prefetch_list) is applied by the backend, not by the modules:
    the lists are offered at every level and size."""
import pytest
import torch

from conftest import TINY_UNET_CFG, load_golden, sub_state_dict
from emu_ops import EmuOps
from tooncrafter_amd import ops


class Recorder(EmuOps):
    """EmuOps that logs, in call order, ("norm", [prefetch tensors]) and ("gemm", weight tensor)."""

    def __init__(self):
        super().__init__(round_bf16=True)
        self.log = []

    def gemm(self, a, w, bias=None, **kw):
        self.log.append(("gemm", w))
        return super().gemm(a, w, bias, **kw)

    def groupnorm(self, x, gamma, beta, *, prefetch=None, **kw):
        self.log.append(("gn", list(prefetch or [])))
        return super().groupnorm(x, gamma, beta, **kw)

    def layernorm(self, x, gamma, beta, eps=1e-5, mx_for=None, prefetch=None):
        self.log.append(("ln", [t for t in (prefetch or []) if t is not None]))
        return super().layernorm(x, gamma, beta, eps, mx_for=mx_for)

    # the SHIPPED composition of norm + convolution (ops.HipOps.gn_conv: it only calls self.groupnorm / self.gemm), so that the
    # list it builds is what gets recorded; the one-launch level-0 operators of the HIP backend do not exist here
    gn_conv = ops.HipOps.gn_conv


@pytest.fixture()
def recorder():
    rec = Recorder()
    prev = ops.set_backend(rec)
    yield rec
    ops.set_backend(prev)


def test_every_prefetched_tensor_is_the_weight_of_a_gemm_that_follows(tiny_sd, recorder):
    from tooncrafter_amd.lvdm.openaimodel3d import UNetModel
    g = load_golden("unet_tiny.npz")
    un = UNetModel(**TINY_UNET_CFG).eval()
    un.load_state_dict(sub_state_dict(tiny_sd, "model.diffusion_model."), strict=True)
    with torch.no_grad():
        un(torch.from_numpy(g["x"]), torch.from_numpy(g["timesteps"]), context=torch.from_numpy(g["context"]),
           fs=torch.from_numpy(g["fs"]))
    log = recorder.log
    norms = [(i, kind, lst) for i, (kind, lst) in enumerate(log) if kind in ("gn", "ln")]
    offered = [n for n in norms if n[2]]
    assert len(norms) > 100 and len(offered) >= 0.8 * len(norms), (len(norms), len(offered))
    first_is_next = 0
    for i, kind, lst in offered:
        # the next GEMMs; a cross-attention's first call also projects its context (K / V of the text and image tokens, cached
        # per conditioning afterwards) between the LayerNorm and the q projection: up to four more GEMMs in this one forward
        following = [w for k, w in log[i + 1:i + 60] if k == "gemm"][:8]
        assert following, f"call {i}: a {kind} with a prefetch list and no GEMM behind it"
        for j, t in enumerate(lst):
            pos = [n for n, w in enumerate(following) if w is t]
            assert pos, f"call {i} ({kind}): prefetch tensor {j} {tuple(t.shape)} is not the weight of any of the next eight GEMMs"
        assert lst[0] is following[0] or (kind == "ln" and any(lst[0] is w for w in following[:6])), \
            f"call {i} ({kind}): the first prefetch tensor should be the consumer right behind the norm"
        first_is_next += lst[0] is following[0]
        assert len(lst) <= 4
    assert first_is_next >= 0.85 * len(offered), (first_is_next, len(offered))
    # both kinds of norm take part, and some lists name two GEMMs (consumer + the one after it)
    assert any(k == "gn" for _, k, _ in offered) and any(k == "ln" for _, k, _ in offered)
    assert any(len(lst) >= 2 for _, _, lst in offered)


def test_prefetch_list_leaves_out_what_the_fp8_route_reads_quantised():
    """with the MXFP8 routing on, a consumer it takes reads a cached quantised copy of W, so the bf16 tensor is
    not worth streaming.  The rule is HipOps.prefetch_list's alone (no GPU needed: it only looks at shapes)."""
    be = ops.HipOps.__new__(ops.HipOps)
    be.prefetch_on, be.prefetch_max_rows, be.prefetch_min_bytes, be.prefetch_max_bytes = True, 8192, 1 << 20, 96 << 20
    be.fp8, be.fp8_min_m, be.fp8_min_k, be.fp8_min_n, be.fp8_n_over_k = None, 1024, 0, 0, 2.0
    be.fp8_min_cin, be.fp8_max_cin = 0, 1280
    wqkv = torch.empty(3840, 1280, dtype=torch.bfloat16)           # N >= 2 K: the fp8 route takes it
    wo = torch.empty(1280, 1280, dtype=torch.bfloat16)             # narrow N: stays bf16
    wg = torch.empty(10240, 1280, dtype=torch.bfloat16)            # GEGLU projection: taken
    conv = torch.empty(1280, 11520, dtype=torch.bfloat16)          # a 3x3 convolution's packed weight
    ids = lambda lst: [id(t) for t in lst]
    assert ids(be.prefetch_list(5120, [wqkv, wo], linear=True)) == ids([wqkv, wo])
    assert ids(be.prefetch_list(5120, [conv])) == ids([conv])
    be.fp8 = "linear"
    assert ids(be.prefetch_list(5120, [wqkv, wo], linear=True)) == ids([wo])
    assert ids(be.prefetch_list(5120, [wg, wo], linear=True)) == ids([wo])
    assert ids(be.prefetch_list(5120, [conv])) == ids([conv])       # "linear" routing leaves the convolutions in bf16
    assert be.prefetch_list(512, [wqkv], linear=True) and True      # below fp8_min_m rows the route declines: streamed
    be.fp8 = "all"
    assert be.prefetch_list(5120, [conv]) == []                    # the convolutions read quantised weights too
    assert ids(be.prefetch_list(5120, [wqkv, wo], linear=True)) == ids([wo])


def test_stale_op_library_is_fatal_not_a_fallback(monkeypatch):
    """an ABI mismatch of libtooncrafter_torch.so must not end in a silent ctypes run."""
    from tooncrafter_amd import _lib, torch_ops
    monkeypatch.setattr(ops, "_backend", None)
    monkeypatch.setattr(torch_ops, "load", lambda: (_ for _ in ()).throw(_lib.TooncrafterAbiError("stale")))
    monkeypatch.delenv("TC_BINDING", raising=False)
    with pytest.raises(_lib.TooncrafterAbiError):
        ops.backend()
    # ... while a MISSING op library is a fallback that says so
    monkeypatch.setattr(torch_ops, "load", lambda: (_ for _ in ()).throw(_lib.TooncrafterHipError("not found")))
    monkeypatch.setattr(ops, "_binding_fallback", None)
    be = ops.backend()
    assert getattr(be, "binding", "ctypes") != "torch" and "not found" in ops.binding_fallback()
    monkeypatch.setattr(ops, "_backend", None)
