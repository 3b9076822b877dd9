"""Micro-benchmark of fx_emb_grad_reduce on the c2 id distribution (26 Criteo-cardinality columns,
B = 4096, power-law ids): HIP-event time of the three launches for D = 16 and D = 1.
usage: python scripts/reduce_bench.py    (DIST=uniform for the no-hot-row case)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/scripts/", 1)[0])
from fuxictr_amd import ops, synthetic  # noqa: E402

DEV = "cuda:0"
B = 4096
cards = synthetic.CRITEO_CARDS
rng = np.random.default_rng(0)
b = synthetic.criteo_batch(rng, B, cards=cards, dist=os.environ.get("DIST", "powerlaw"))
ids = np.stack([b["C%d" % (c + 1)] for c in range(26)], 1).astype(np.int32)
vocabs = [c + 1 for c in cards]
bases = np.concatenate([[0], np.cumsum(vocabs)[:-1]]).astype(np.int64)
R = int(sum(vocabs))
d_ids = torch.from_numpy(ids).to(DEV)
ws = torch.empty(ops.dedup_workspace_bytes(B * 26), dtype=torch.uint8, device=DEV)
dd = ops.dedup(d_ids, torch.from_numpy(bases).to(DEV), torch.tensor(vocabs, dtype=torch.int32, device=DEV),
               torch.zeros(26, dtype=torch.int32, device=DEV), R, ws, columns_sorted=True)
nu = int(dd.n_unique.item())
runs = (dd.seg_start[1:nu + 1] - dd.seg_start[:nu]).cpu().numpy()
print("unique rows %d, runs > 32: %d, longest %d" % (nu, int((runs > 32).sum()), int(runs.max())))
for D in (16, 1):
    n_slots = 39
    dout = torch.randn(B, n_slots * D, device=DEV)
    offs = torch.tensor([(c + 13) * D for c in range(26)], dtype=torch.int64, device=DEV)
    G = torch.zeros(dd.n_max, D, device=DEV)
    sq = torch.empty(ops.emb_grad_reduce_partials(dd.n_max, D), device=DEV)
    scr = torch.zeros(ops.emb_grad_reduce_scratch_ints(dd.n_max), dtype=torch.int32, device=DEV)
    for _ in range(5):
        ops.emb_grad_reduce(dout, n_slots * D, offs, 26, D, dd, G, sq, scr)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 200
    e0.record()
    for _ in range(n):
        ops.emb_grad_reduce(dout, n_slots * D, offs, 26, D, dd, G, sq, scr)
    e1.record()
    torch.cuda.synchronize()
    print("D=%d: %.2f us per call (3 launches)" % (D, e0.elapsed_time(e1) * 1e3 / n))
