import time, torch
dev = torch.device("cuda:0")
h = torch.empty(1 << 18, dtype=torch.float32, pin_memory=True)
x = torch.randn(4096, 4096, device=dev)
for label, busy in (("idle GPU", False), ("GPU busy with later work", True)):
    ts = []
    for _ in range(20):
        d = h.to(dev, non_blocking=True)
        ev = torch.cuda.Event(); ev.record()
        time.sleep(0.003)                       # the copy is long done
        if busy:
            for _ in range(30):
                y = x @ x                       # ~1 ms each, enqueued AFTER the event
        t0 = time.perf_counter(); ev.synchronize(); ts.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
    print("%-28s Event.synchronize of a completed event: median %.3f ms  max %.3f ms" %
          (label, 1e3 * sorted(ts)[len(ts) // 2], 1e3 * max(ts)))
    ts = []
    for _ in range(20):
        d = h.to(dev, non_blocking=True)
        ev = torch.cuda.Event(); ev.record()
        time.sleep(0.003)
        if busy:
            for _ in range(30):
                y = x @ x
        t0 = time.perf_counter(); q = ev.query(); ts.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
    print("%-28s Event.query:                              median %.3f ms (returned %s)" %
          (label, 1e3 * sorted(ts)[len(ts) // 2], q))
