#!/usr/bin/env python3
"""Micro-benchmark of fx_gemm_f32 on the shapes of the DeepFM / DCNv2 towers (B = 4096)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fuxictr_amd import ops  # noqa: E402

dev = "cuda:0"
B = int(os.environ.get("FX_B", 4096))
SHAPES = [  # name, transa, transb, M, N, K, split_k
    ("fwd 624->1024", 0, 1, B, 1024, 624, 1),
    ("fwd 1024->1024", 0, 1, B, 1024, 1024, 1),
    ("fwd 1024->1", 0, 1, B, 1, 1024, 1),
    ("cross 624->624", 0, 1, B, 624, 624, 1),
    ("dX 1024->1024", 0, 0, B, 1024, 1024, 1),
    ("dX 1024->624", 0, 0, B, 624, 1024, 1),
    ("dX 1->1024", 0, 0, B, 1024, 1, 1),
    ("dW 1024x1024", 1, 0, 1024, 1024, B, 4),
    ("dW 1024x624", 1, 0, 1024, 624, B, 6),
    ("dW 624x624", 1, 0, 624, 624, B, 10),
    ("dW 1x1024", 1, 0, 1, 1024, B, 64),
    ("square 4096", 0, 1, 4096, 4096, 4096, 1),
]
g = torch.Generator(device=dev).manual_seed(0)
out = []
for name, ta, tb, M, N, K, sk in SHAPES:
    A = torch.randn((K, M) if ta else (M, K), device=dev, generator=g)
    Bm = torch.randn((N, K) if tb else (K, N), device=dev, generator=g)
    C = torch.empty(M, N, device=dev)
    ws = torch.empty(max(sk * M * N, 1), device=dev)
    for _ in range(3):
        ops.gemm(A, Bm, C, transa=bool(ta), transb=bool(tb), split_k=sk, workspace=ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        ops.gemm(A, Bm, C, transa=bool(ta), transb=bool(tb), split_k=sk, workspace=ws)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    tf = 2.0 * M * N * K / (us * 1e-6) / 1e12
    out.append({"shape": name, "us": round(us, 2), "TFLOPs": round(tf, 2)})
    print("%-18s %9.2f us  %7.2f TFLOP/s" % (name, us, tf), flush=True)
print(json.dumps(out))
