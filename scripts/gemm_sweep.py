#!/usr/bin/env python3
"""K / M sweeps of fx_gemm_f32 (NT layout) to separate fixed per-launch cost from per-k-tile cost."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fuxictr_amd import ops  # noqa: E402

dev = "cuda:0"
only = os.environ.get("FX_ONLY")


def run(M, N, K, n=20):
    A = torch.randn(M, K, device=dev)
    Bm = torch.randn(N, K, device=dev)
    C = torch.empty(M, N, device=dev)
    for _ in range(3):
        ops.gemm(A, Bm, C, transb=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ops.gemm(A, Bm, C, transb=True)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print("M=%5d N=%5d K=%5d  %9.2f us  %7.2f TFLOP/s" % (M, N, K, us, 2.0 * M * N * K / us / 1e6),
          flush=True)


if only:
    M, N, K = [int(x) for x in only.split(",")]
    run(M, N, K, n=10)
else:
    for K in [32, 128, 256, 512, 1024, 2048, 4096, 8192]:
        run(4096, 1024, K)
    for M in [1024, 2048, 4096, 8192, 16384, 32768]:
        run(M, 1024, 1024)
    for N in [512, 1024, 2048, 4096]:
        run(4096, N, 1024)
