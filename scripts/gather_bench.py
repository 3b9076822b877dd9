#!/usr/bin/env python3
"""HBM roofline of fx_emb_gather_fwd: full Criteo tables (33.76 M rows x 16 fp32 = 2.16 GB), 26 id
columns + 13 numeric, batch swept from the training batch (4096: one HBM latency of data in flight)
to 512 K (bandwidth regime).  Bytes per sample: 1820 in (26 rows of 64 B + ids + numerics, SURVEY.md
8d) + 2496 out (the [39,16] record)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from fuxictr_amd import ops, synthetic  # noqa: E402

dev = torch.device("cuda:0")
cards = synthetic.CRITEO_CARDS
D, Fd = 16, 13
vocab = [c + 1 for c in cards]
base = np.concatenate([[0], np.cumsum(vocab)[:-1]])
table = torch.randn(sum(vocab), D, device=dev)
row_base = torch.tensor(base, dtype=torch.int64, device=dev)
col_vocab = torch.tensor(vocab, dtype=torch.int32, device=dev)
C = len(cards)
out_off = torch.arange(Fd, Fd + C, dtype=torch.int64, device=dev) * D
num_off = torch.arange(0, Fd, dtype=torch.int64, device=dev) * D
num_w = torch.randn(Fd, D, device=dev)
scal = ops.new_scalars(dev)
res = []
for dist in ("uniform", "powerlaw"):
    for B in (4096, 32768, 131072, 524288):
        rng = np.random.default_rng(B)
        b = synthetic.criteo_batch(rng, B, dist=dist)
        ids = torch.from_numpy(np.stack([b["C%d" % (i + 1)] for i in range(C)], 1)).to(dev).int()
        dense = torch.from_numpy(np.stack([b["I%d" % (i + 1)] for i in range(Fd)], 1)).to(dev).float()
        out = torch.empty(B, (C + Fd) * D, device=dev)
        for _ in range(3):
            ops.emb_gather_fwd(table, D, ids, row_base, col_vocab, out_off, dense, num_w, num_off, out, scal)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            ops.emb_gather_fwd(table, D, ids, row_base, col_vocab, out_off, dense, num_w, num_off, out, scal)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        b_in, b_out = B * (C * (4 * D + 4) + Fd * 4), B * (C + Fd) * 4 * D
        r = {"dist": dist, "B": B, "us": round(us, 2), "GBps_in": round(b_in / us / 1e3, 1),
             "GBps_in_out": round((b_in + b_out) / us / 1e3, 1),
             "frac_of_8TBps": round((b_in + b_out) / us / 1e3 / 8000.0, 3)}
        res.append(r)
        print(r, flush=True)
print(json.dumps(res))
