#!/usr/bin/env python3
""""a second number *with* the reference DataLoader".  Host side of that number, measurable
without a GPU: how many samples/s the reference's own RankDataLoader -> NpzDataLoader (npz_dataloader.py:35-125:
per-row __getitem__, default_collate, BatchCollator's column slicing into a dict of tensors) hands out on this
host for the c2 Criteo shape at batch 4096 — the ceiling of ANY model behind it.  Needs a reference checkout
(FX_REFERENCE_ROOT or /root/reference); builder-side only, nothing in tests / bench reads it.
usage: ref_loader_rate.py [--workers 0,3,8] [--batches 40]"""
import argparse
import os
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", default="0,3,8")
    ap.add_argument("--batches", type=int, default=40)
    ap.add_argument("--batch", type=int, default=4096)
    args = ap.parse_args()
    ref = os.environ.get("FX_REFERENCE_ROOT", "/root/reference")
    for name in ["polars", "h5py", "keras_preprocessing", "keras_preprocessing.sequence"]:
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["keras_preprocessing.sequence"].pad_sequences = lambda *a, **k: None
    sys.dont_write_bytecode = True
    sys.path.insert(0, ref)
    import torch
    from fuxictr.pytorch.dataloaders import RankDataLoader
    from fuxictr_amd import synthetic
    fmap, spec = synthetic.criteo_feature_map(embedding_dim=16)
    rng = np.random.default_rng(7)
    n_rows = args.batch * (args.batches + 8)
    big = synthetic.criteo_batch(rng, n_rows, dist="powerlaw")
    path = "/tmp/fx_ref_loader_rate.npz"
    np.savez(path, **{k: np.asarray(v) for k, v in big.items()})
    print("host cores %d, torch threads %d, rows %d, batch %d" % (len(os.sched_getaffinity(0)),
                                                                  torch.get_num_threads(), n_rows, args.batch))
    for w in [int(x) for x in args.workers.split(",")]:
        t0 = time.perf_counter()
        gen, _ = RankDataLoader(fmap, stage="train", train_data=path, batch_size=args.batch, shuffle=True,
                                num_workers=w).make_iterator()
        t_build = time.perf_counter() - t0
        it = iter(gen)
        for _ in range(4):                       # worker start-up, first batches
            b = next(it)
        t0 = time.perf_counter()
        n = 0
        for _ in range(args.batches):
            b = next(it)
            n += next(iter(b.values())).shape[0]
        dt = time.perf_counter() - t0
        dts = sorted({str(v.dtype) for v in b.values()})
        print("num_workers %d: %8.0f samples/s (%.1f ms per batch of %d; loader built in %.1f s; batch = dict of "
              "%d tensors, dtypes %s)" % (w, n / dt, 1e3 * dt / args.batches, args.batch, t_build, len(b), dts))
        del it, gen


if __name__ == "__main__":
    main()
