#!/usr/bin/env python3
"""GEMM probe for the tower shapes (B = 4096): K sweep (fixed cost vs slope), split-K choices of the
weight gradients, epilogue cost.  Back-to-back launches between one event pair (kernel + ~1.3 us
boundary).  Tile forcing (FX_GEMM_TILE, FX_GEMM_W64) is read once per process: run several times."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fuxictr_amd import ops  # noqa: E402

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
tag = "tile=%s w64=%s" % (os.environ.get("FX_GEMM_TILE", "auto"), os.environ.get("FX_GEMM_W64", "4"))


def run(name, ta, tb, M, N, K, sk=1, bias=False, act=0, mask=False, add=False, rowsum=False, n=20):
    A = torch.randn((K, M) if ta else (M, K), device=dev, generator=g)
    Bm = torch.randn((N, K) if tb else (K, N), device=dev, generator=g)
    C = torch.empty(M, N, device=dev)
    ws = torch.empty(ops.gemm_workspace_floats(M, N, sk), device=dev)
    kw = dict(transa=bool(ta), transb=bool(tb), split_k=sk, workspace=ws)
    if bias:
        kw["bias"] = torch.randn(N, device=dev, generator=g)
    if act:
        kw["act"] = 1
    if mask:
        kw["mask"] = torch.randn(M, N, device=dev, generator=g)
    if add:
        kw["add"] = torch.randn(M, N, device=dev, generator=g)
    if rowsum:
        kw["rowsum"] = torch.empty(M, device=dev)
    for _ in range(3):
        ops.gemm(A, Bm, C, **kw)
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(int(2e6))
        e0.record()
        for _ in range(n):
            ops.gemm(A, Bm, C, **kw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        best = us if best is None else min(best, us)
    tf = 2.0 * M * N * K / (best * 1e-6) / 1e12
    print("[%s] %-34s %8.2f us  %7.2f TF  %.3f" % (tag, name, best, tf, tf / 157.3), flush=True)
    return best


B = 4096
print("-- K sweep, fwd layout (A[M,K] x W[N,K]^T + bias + relu), M=4096 N=1024")
for K in (64, 128, 256, 512, 1024, 2048):
    run("fwd 4096x1024 K=%d" % K, 0, 1, B, 1024, K, bias=True, act=1)
print("-- epilogue cost at K=1024")
run("fwd plain", 0, 1, B, 1024, 1024)
run("fwd bias+relu", 0, 1, B, 1024, 1024, bias=True, act=1)
run("dX plain", 0, 0, B, 1024, 1024)
run("dX + relu mask", 0, 0, B, 1024, 1024, mask=True)
print("-- weight gradients 1024x1024x4096 (dz^T h), split-K + slab reduce (+ fused bias gradient)")
for sk in (1, 2, 4, 8):
    run("dW 1024x1024 split_k=%d rowsum" % sk, 1, 0, 1024, 1024, B, sk=sk, rowsum=True)
for sk in (1, 2, 4):
    run("dW 1024x624 split_k=%d rowsum" % sk, 1, 0, 1024, 624, B, sk=sk, rowsum=True)
print("-- CrossNet shapes")
run("cross fwd 4096x624x624 (bias+add)", 0, 1, B, 624, 624, bias=True, add=True)
run("cross dX 4096x624x624 (add)", 0, 0, B, 624, 624, add=True)
for sk in (1, 2, 4, 8):
    run("cross dW 624x624x4096 split_k=%d" % sk, 1, 0, 624, 624, B, sk=sk, rowsum=True)
