import time, numpy as np, torch
dev = torch.device("cuda:0")
n = 26 * 4096
cols = [torch.from_numpy(np.random.randint(0, 1000, 4096)) for _ in range(26)]
pin = [torch.empty(n, dtype=torch.int64, pin_memory=True) for _ in range(2)]
x = torch.randn(2048, 2048, device=dev)
T = {}
def tick(name, t0):
    T[name] = T.get(name, 0.0) + time.perf_counter() - t0
evs = [None, None]
for it in range(60):
    s = it & 1
    t0 = time.perf_counter()
    if evs[s] is not None: evs[s].synchronize()
    tick("sync", t0); t0 = time.perf_counter()
    c = torch.cat(cols)
    tick("cat", t0); t0 = time.perf_counter()
    pin[s].copy_(c)
    tick("to_pinned", t0); t0 = time.perf_counter()
    d = pin[s].to(dev, non_blocking=True)
    tick("h2d_enqueue", t0); t0 = time.perf_counter()
    ev = torch.cuda.Event(); ev.record(); evs[s] = ev
    tick("record", t0); t0 = time.perf_counter()
    for _ in range(12): y = x @ x            # ~1.3 ms of GPU work per "step"
    tick("launch_work", t0)
torch.cuda.synchronize()
for k, v in T.items(): print("%-12s %.3f ms/iter" % (k, 1e3 * v / 60))
T.clear()
for it in range(60):
    t0 = time.perf_counter()
    d = torch.cat(cols).to(dev)                      # pageable source: the driver stages it
    tick("cat+pageable_to", t0); t0 = time.perf_counter()
    for _ in range(12): y = x @ x
    tick("launch_work", t0)
torch.cuda.synchronize()
for k, v in T.items(): print("%-16s %.3f ms/iter" % (k, 1e3 * v / 60))
T.clear()
import ctypes
for it in range(60):
    t0 = time.perf_counter()
    c = torch.cat(cols)
    ctypes.memmove(pin[0].data_ptr(), c.data_ptr(), c.numel() * 8)
    tick("memmove_to_pinned", t0)
for k, v in T.items(): print("%-16s %.3f ms/iter" % (k, 1e3 * v / 60))
T.clear()
pn = pin[0].numpy()
for it in range(60):
    t0 = time.perf_counter()
    np.concatenate([c.numpy() for c in cols], out=pn)
    tick("np.concatenate_out_pinned", t0)
for k, v in T.items(): print("%-16s %.3f ms/iter" % (k, 1e3 * v / 60))
