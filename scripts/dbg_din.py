"""GPU debugging aid: the generic de-dup at c4's size inside a captured hipGraph, against numpy."""
import faulthandler
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.enable()
from fuxictr_amd import ops  # noqa: E402

DEV = torch.device("cuda:0")


def graph_case(B, C, vocab, reps=6):
    rng = np.random.default_rng(B + C)
    vocabs = [vocab] * C
    bases = np.zeros(C, dtype=np.int64)
    ws = torch.empty(ops.dedup_workspace_bytes(B * C), dtype=torch.uint8, device=DEV)
    ids_dev = torch.zeros(B, C, dtype=torch.int32, device=DEV)
    bases_d = torch.from_numpy(bases).to(DEV)
    vocab_d = torch.tensor(vocabs, dtype=torch.int32, device=DEV)
    pad_d = torch.zeros(C, dtype=torch.int32, device=DEV)

    def draw():
        ids = np.minimum((vocab * rng.random((B, C)) ** 3).astype(np.int64), vocab - 1)
        ids[rng.random(ids.shape) < 0.3] = 0
        return ids

    def run():
        return ops.dedup(ids_dev, bases_d, vocab_d, pad_d, vocab, ws, want_uid=True)

    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):
            ids_dev.copy_(torch.from_numpy(draw().astype(np.int32)))
            run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        dd = run()
    for r in range(reps):
        ids = draw()
        ids_dev.copy_(torch.from_numpy(ids.astype(np.int32)))
        g.replay()
        torch.cuda.synchronize()
        keys = ids.reshape(-1)
        uniq, counts = np.unique(keys[keys != 0], return_counts=True)
        nu = int(dd.n_unique.item())
        got = dd.uniq_row[:nu].cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        seg = dd.seg_start[:nu + 1].cpu().numpy().astype(np.int64)
        ok = nu == len(uniq) and np.array_equal(got, uniq) and np.array_equal(np.diff(seg), counts)
        print("graph replay %d  B=%d C=%d: n_unique %d (numpy %d) %s" % (
            r, B, C, nu, len(uniq), "OK" if ok else "MISMATCH"), flush=True)


graph_case(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))
