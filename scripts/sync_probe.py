"""What does the bracket of the timed region cost?  20 replays of a captured DeepFM-sized chain of kernels, closed by
(A) torch.cuda.synchronize()  (B) stream.synchronize() then torch.cuda.synchronize()  (C) stream.synchronize() only;
wall clock against the HIP events around the same replays."""
import time
import torch

dev = torch.device("cuda:0")
x = torch.randn(4096, 1024, device=dev)
w = [torch.randn(1024, 1024, device=dev) for _ in range(12)]
side = torch.cuda.Stream()
extra = [torch.cuda.Stream() for _ in range(4)]        # (a process with several streams, as the bench has)
for s in extra:
    with torch.cuda.stream(s):
        torch.zeros(8, device=dev).add_(1)
torch.cuda.synchronize()
with torch.cuda.stream(side):
    y = x
    for wi in w:
        y = torch.relu(y @ wi) * 1e-2
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        y = x
        for wi in w:
            y = torch.relu(y @ wi) * 1e-2
torch.cuda.synchronize()
for mode in ("A device", "B stream+device", "C stream", "A device", "B stream+device", "C stream"):
    for _ in range(30):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(20):
        g.replay()
    e1.record()
    if mode[0] in "BC":
        torch.cuda.current_stream().synchronize()
    t1 = time.perf_counter()
    if mode[0] in "AB":
        torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-16s wall %.3f ms (after stream sync %.3f)  events %.3f ms  wall - events %.3f ms" % (
        mode, 1e3 * (t2 - t0), 1e3 * (t1 - t0), e0.elapsed_time(e1), 1e3 * (t2 - t0) - e0.elapsed_time(e1)))
