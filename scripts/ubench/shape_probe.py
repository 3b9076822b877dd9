import sys
sys.path.insert(0, "/root/repo")
import torch
from fuxictr_amd import ops
dev = "cuda:0"
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for (M, N, K, ta, tb, sk) in [(4096, 624, 624, 0, 1, 1), (4096, 640, 640, 0, 1, 1), (4096, 624, 624, 0, 0, 1), (4096, 640, 640, 0, 0, 1),
                              (624, 624, 4096, 1, 0, 6), (640, 640, 4096, 1, 0, 6), (4096, 1024, 624, 0, 1, 1), (4096, 1024, 640, 0, 1, 1)]:
    A = torch.randn((K, M) if ta else (M, K), device=dev); B = torch.randn((N, K) if tb else (K, N), device=dev)
    C = torch.empty(M, N, device=dev); ws = torch.empty(sk * M * N + 8192, device=dev)
    t = timeit(lambda: ops.gemm(A, B, C, transa=bool(ta), transb=bool(tb), split_k=sk, workspace=ws))
    print("M=%d N=%d K=%d ta=%d tb=%d sk=%d: %6.1f us  %5.1f TF" % (M, N, K, ta, tb, sk, t, 2.0 * M * N * K / t / 1e6), flush=True)
