#!/usr/bin/env python3
"""Yardstick: fx_gemm_f32 vs torch.mm (rocBLAS / hipBLASLt fp32) on the tower shapes, B = 4096."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fuxictr_amd import ops  # noqa: E402

dev = "cuda:0"
torch.backends.cuda.matmul.allow_tf32 = False
B = 4096
SHAPES = [("fwd 1024->1024", 0, 1, B, 1024, 1024), ("fwd 624->1024", 0, 1, B, 1024, 624),
          ("dX 1024->1024", 0, 0, B, 1024, 1024), ("dW 1024x1024", 1, 0, 1024, 1024, B),
          ("cross 624->624", 0, 1, B, 624, 624), ("square 4096", 0, 1, 4096, 4096, 4096)]


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


g = torch.Generator(device=dev).manual_seed(0)
for name, ta, tb, M, N, K in SHAPES:
    A = torch.randn((K, M) if ta else (M, K), device=dev, generator=g)
    Bm = torch.randn((N, K) if tb else (K, N), device=dev, generator=g)
    C = torch.empty(M, N, device=dev)
    sk = 4 if ta else 1
    ws = torch.empty(max(sk * M * N, 1), device=dev)
    t_fx = timeit(lambda: ops.gemm(A, Bm, C, transa=bool(ta), transb=bool(tb), split_k=sk, workspace=ws))
    Aop = A.t() if ta else A
    Bop = Bm.t() if tb else Bm
    t_bl = timeit(lambda: torch.mm(Aop, Bop, out=C))
    fl = 2.0 * M * N * K / 1e6
    print("%-16s fx %7.1f us %6.1f TF | torch.mm %7.1f us %6.1f TF" % (name, t_fx, fl / t_fx, t_bl, fl / t_bl),
          flush=True)
