#!/usr/bin/env python3
"""What the per-step `loss.item()` of the reference's train_epoch (rank_model.py:333) costs on the native step:
200 replayed DeepFM steps (c2 shape) read back every step vs accumulated on the device (train_epoch, round 4)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    sys.argv = [sys.argv[0]]
    args = bench.parse()
    from fuxictr_amd import synthetic
    cards = [max(3, int(c * args.vocab_scale)) for c in synthetic.CRITEO_CARDS]
    model, fmap, spec = bench.build_model(args, 0, cards)
    dev = torch.device("cuda:0")
    pool = bench.make_pool(args, 0, cards, spec, dev, 64)
    model.train()
    for i in range(30):
        model.train_step(pool[i % 64])
    torch.cuda.synchronize()
    for mode in ("item every step", "device sum", "item every step", "device sum"):
        acc = torch.zeros((), dtype=torch.float64, device=dev)
        host = 0.0
        t0 = time.perf_counter()
        for i in range(200):
            loss = model.train_step(pool[i % 64])
            if mode.startswith("item"):
                host += loss.item()
            else:
                acc.add_(loss.detach())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%-16s %.4f ms/step" % (mode, 1e3 * dt / 200))


if __name__ == "__main__":
    main()
