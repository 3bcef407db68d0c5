#!/usr/bin/env python3
"""GPU box health check with PLAIN PyTorch only -- libtooncrafter_hip.so is never loaded here.
Round-1's driver run and several round-2 runs died with 'Memory access fault by GPU node-2' on one particular GPU
of the pool while the same commands pass on every other lease; this script separates "the box is unhealthy" from
"our kernels are wrong": H2D/D2H copies with verification, elementwise kernels, bf16 GEMMs, a hipGraph replay.
Exit code 0 = healthy.  A GPU memory fault aborts the process (exit 134) -- run it as a subprocess."""
import sys
import time

import torch


def main():
    assert torch.cuda.is_available(), "no GPU visible"
    dev = torch.device("cuda:0")
    t0 = time.time()
    props = torch.cuda.get_device_properties(0)
    print(f"[health] {props.name}, {props.total_memory / 2**30:.0f} GiB, {props.multi_processor_count} CUs", flush=True)
    g = torch.Generator().manual_seed(0)
    for n in (1, 1000, 1 << 16, 1 << 22, 1 << 26):                      # H2D / D2H round trips, pageable memory
        x = torch.randn(n, generator=g)
        y = x.to(dev)
        assert torch.equal(y.cpu(), x), f"H2D/D2H mismatch at n={n}"
    print("[health] copies ok", flush=True)
    a = torch.randn(1 << 26, device=dev)
    b = (a * 2.0 + 1.0).sin().sum()
    torch.cuda.synchronize()
    assert torch.isfinite(b)
    print("[health] elementwise ok", flush=True)
    m = torch.randn(8192, 8192, device=dev).to(torch.bfloat16)
    acc = None
    for _ in range(20):
        acc = m @ m
    torch.cuda.synchronize()
    assert torch.isfinite(acc.float()).all()
    print("[health] bf16 GEMM ok", flush=True)
    big = torch.empty(64 << 30, dtype=torch.uint8, device=dev)          # touch 64 GiB of HBM
    big.fill_(3)
    assert int(big[::1 << 20].sum()) == 3 * (64 << 10)
    del big
    print("[health] 64 GiB fill ok", flush=True)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    x = torch.randn(4096, 4096, device=dev)
    with torch.cuda.stream(s):
        x @ x
    torch.cuda.current_stream().wait_stream(s)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        y = (x @ x).relu()
    for _ in range(10):
        gr.replay()
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()
    print("[health] hipGraph replay ok", flush=True)
    # a few seconds of what the bench looks like to the box: sustained bf16 GEMMs at full power, allocator churn
    # (hundreds of 50-700 MB tensors created and freed), host<->device traffic in between
    m = torch.randn(8192, 8192, device=dev).to(torch.bfloat16)
    t_end = time.time() + 4.0
    n = 0
    while time.time() < t_end:
        bufs = [torch.empty((81920 * (1 + i % 4), 320), device=dev, dtype=torch.bfloat16).normal_() for i in range(6)]
        for b in bufs:
            acc = m @ m
            b.mul_(0.5)
        h = bufs[0][:4096].cpu()
        assert torch.isfinite(h.float()).all()
        del bufs
        n += 1
    torch.cuda.synchronize()
    assert torch.isfinite(acc.float()).all()
    print(f"[health] sustained load ok ({n} rounds); all checks passed in {time.time() - t0:.1f} s", flush=True)


if __name__ == "__main__":
    main()
    sys.exit(0)
