import os, sys
sys.path.insert(0, "/root/repo")
import torch
from fuxictr_amd import ops
dev = "cuda:0"
M, N = 4096, 1024
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for K in [128, 256, 512, 1024, 2048, 4096, 8192]:
    A = (torch.ones(M, K, device=dev) if os.environ.get("ONES") else torch.randn(M, K, device=dev)); B = (torch.ones(N, K, device=dev) if os.environ.get("ONES") else torch.randn(N, K, device=dev)); C = torch.empty(M, N, device=dev)
    t = timeit(lambda: ops.gemm(A, B, C, transa=False, transb=True))
    tb = timeit(lambda: torch.mm(A, B.t(), out=C))
    print("K=%5d  fx %7.1f us (%5.1f TF)  blas %7.1f us (%5.1f TF)" % (K, t, 2.0*M*N*K/t/1e6, tb, 2.0*M*N*K/tb/1e6), flush=True)
