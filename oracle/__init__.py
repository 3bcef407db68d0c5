"""CPU oracle for the ToonCrafter denoising hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain-PyTorch fp32 restatement of
the reference's algorithm (Doubiiu/ToonCrafter, `lvdm/`): DDIM sampler, v-param
algebra, spatio-temporal UNet, dual-reference VideoDecoder.  It exists so the
hand-written HIP path can be checked on boxes where `/root/reference` does not
exist.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it; the product package `tooncrafter_amd` never does.

Pinning: the reference ships no tests or golden vectors,
so the oracle is pinned against outputs of the reference itself, generated in
the build container by `tests/golden/make_golden.py` (which imports the
read-only reference with import stubs) and committed as `tests/golden/*.npz`.
`tests/test_oracle_golden.py` replays them.  Arithmetic living in third-party
kernels (ATen conv/GEMM/norm, xformers attention) is pinned only through those
reference runs.

Every function cites the reference file:line it restates.
"""
