#!/usr/bin/env python3
"""Does running the dX and dW GEMMs of one layer on two HIP streams beat running them back to back?
(they are independent given dZ; each alone fills the chip with ~2 workgroups per CU)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fuxictr_amd import ops  # noqa: E402

dev = "cuda:0"
B, H = 4096, 1024
g = torch.Generator(device=dev).manual_seed(0)
dZ = torch.randn(B, H, device=dev, generator=g)
W = torch.randn(H, H, device=dev, generator=g)
X = torch.randn(B, H, device=dev, generator=g)
dX = torch.empty(B, H, device=dev)
dW = torch.empty(H, H, device=dev)
ws = torch.empty(4 * H * H, device=dev)
side = torch.cuda.Stream()


def serial():
    ops.gemm(dZ, W, dX, transa=False, transb=False)
    ops.gemm(dZ, X, dW, transa=True, transb=False, split_k=4, workspace=ws)


def overlapped():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        ops.gemm(dZ, X, dW, transa=True, transb=False, split_k=4, workspace=ws)
    ops.gemm(dZ, W, dX, transa=False, transb=False)
    main.wait_stream(side)


def timeit(fn, n=50):
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


def graphed(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(4):
                fn()
    return lambda: gr.replay()


fl = 2 * 2.0 * B * H * H
for name, fn in [("serial eager", serial), ("2-stream eager", overlapped)]:
    us = timeit(fn)
    print("%-22s %8.2f us per (dX+dW)  %6.1f TFLOP/s" % (name, us, fl / us / 1e6), flush=True)
for name, fn in [("serial graph", serial), ("2-stream graph", overlapped)]:
    rp = graphed(fn)
    us = timeit(rp) / 4
    print("%-22s %8.2f us per (dX+dW)  %6.1f TFLOP/s" % (name, us, fl / us / 1e6), flush=True)
