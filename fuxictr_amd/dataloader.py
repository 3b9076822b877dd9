"""Device-side batch pipeline.

Drop-in for the reference's `NpzDataLoader` through the hook `RankDataLoader` already has
(`data_loader=` kwarg, fuxictr/pytorch/dataloaders/rank_dataloader.py:51-52):

    RankDataLoader(feature_map, stage="train", data_loader=DeviceNpzDataLoader, **params)

The reference stacks every column into ONE float64 matrix, fetches a batch row by row through
`Dataset.__getitem__` + `default_collate` (npz_dataloader.py:35-66, :101-125) and the model then
copies 40 columns to the device one at a time (rank_model.py:186).  Here the columns are kept
column-major in two host blocks (ids int32, numerics + labels fp32); a batch is one `np.take` per
column into a pinned staging block and ONE asynchronous H2D copy per block on a copy stream, prepared
by a background thread while the model trains on the previous batch.  The yielded dict maps column
names to contiguous device views, so `BaseModel.get_inputs` has nothing left to copy.  A batch's
views are valid until the next batch is requested from the iterator (the ring slot is then reused).
"""
import queue
import threading

import numpy as np
import torch


class DeviceNpzDataLoader(object):
    def __init__(self, feature_map, data_path, split="train", batch_size=32, shuffle=False,
                 device=None, prefetch=2, seed=None, **kwargs):
        if not data_path.endswith(".npz"):
            data_path += ".npz"
        self.feature_map = feature_map
        self.batch_size = int(batch_size)
        self.shuffle = bool(shuffle)
        self.device = torch.device(device) if device is not None else (
            torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available()
            else torch.device("cpu"))
        self.prefetch = max(1, int(prefetch))
        self._rng = np.random.default_rng(seed)
        data = np.load(data_path)
        self._id_cols, self._f_cols = [], []      # (name, first row in the block, width)
        id_rows, f_rows = [], []
        for name, spec in feature_map.features.items():
            arr = data[name]
            if spec["type"] == "meta":
                continue
            if spec["type"] in ("categorical", "sequence"):
                a2 = arr.reshape(arr.shape[0], -1).astype(np.int32, copy=False)
                self._id_cols.append((name, len(id_rows), a2.shape[1], arr.ndim > 1))
                id_rows.extend(np.ascontiguousarray(a2[:, j]) for j in range(a2.shape[1]))
            else:
                self._f_cols.append((name, len(f_rows), 1, False))
                f_rows.append(np.ascontiguousarray(arr.reshape(-1).astype(np.float32)))
        for name in feature_map.labels:
            self._f_cols.append((name, len(f_rows), 1, False))
            f_rows.append(np.ascontiguousarray(data[name].reshape(-1).astype(np.float32)))
        self._ids = id_rows                       # column-major: one contiguous array per column
        self._floats = f_rows
        self.num_samples = int(f_rows[0].shape[0])
        self.num_blocks = 1
        self.num_batches = int(np.ceil(self.num_samples / float(self.batch_size)))

    def __len__(self):
        return self.num_batches

    # -- one batch -------------------------------------------------------------------------------
    def _stage(self, idx, bufs):
        n = len(idx)
        ids_h, f_h = bufs
        for c, col in enumerate(self._ids):
            np.take(col, idx, out=ids_h[c, :n])
        for c, col in enumerate(self._floats):
            np.take(col, idx, out=f_h[c, :n])
        return n

    def _views(self, ids_d, f_d, n):
        out = {}
        for name, r0, w, is_seq in self._id_cols:
            out[name] = ids_d[r0, :n] if not is_seq else ids_d[r0:r0 + w, :n].t()
        for name, r0, _, _ in self._f_cols:
            out[name] = f_d[r0, :n]
        return out

    def __iter__(self):
        order = self._rng.permutation(self.num_samples) if self.shuffle \
            else np.arange(self.num_samples)
        B = self.batch_size
        chunks = [order[i:i + B] for i in range(0, self.num_samples, B)]
        if self.device.type != "cuda":
            for idx in chunks:
                ids_h = np.empty((max(len(self._ids), 1), B), dtype=np.int32)
                f_h = np.empty((len(self._floats), B), dtype=np.float32)
                n = self._stage(idx, (ids_h, f_h))
                yield self._views(torch.from_numpy(ids_h), torch.from_numpy(f_h), n)
            return
        # pinned ring + copy stream; the producer thread stays `prefetch` batches ahead
        depth = self.prefetch + 1
        ring = []
        for _ in range(depth):
            ids_t = torch.empty(max(len(self._ids), 1), B, dtype=torch.int32, pin_memory=True)
            f_t = torch.empty(len(self._floats), B, dtype=torch.float32, pin_memory=True)
            ring.append((ids_t, f_t, ids_t.numpy(), f_t.numpy(),
                         torch.empty_like(ids_t, device=self.device),
                         torch.empty_like(f_t, device=self.device), threading.Event()))
            ring[-1][6].set()                      # slot free
        last_copy = [None] * depth                 # the H2D copy that last read a slot's pinned block
        q = queue.Queue(maxsize=self.prefetch)
        copy_stream = torch.cuda.Stream(self.device)
        stop = threading.Event()

        def produce():
            torch.cuda.set_device(self.device)
            for i, idx in enumerate(chunks):
                slot = ring[i % depth]
                while not slot[6].wait(0.05):
                    if stop.is_set():
                        return
                slot[6].clear()
                if last_copy[i % depth] is not None:
                    last_copy[i % depth].synchronize()   # the pinned block is about to be rewritten
                n = self._stage(idx, (slot[2], slot[3]))
                with torch.cuda.stream(copy_stream):
                    slot[4].copy_(slot[0], non_blocking=True)
                    slot[5].copy_(slot[1], non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(copy_stream)
                last_copy[i % depth] = ev
                q.put((i % depth, n, ev))
            q.put(None)

        th = threading.Thread(target=produce, daemon=True)
        th.start()
        prev = None
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                si, n, ev = item
                torch.cuda.current_stream(self.device).wait_event(ev)
                if prev is not None:
                    # the consumer is done issuing work on the previous batch: its slot may be
                    # overwritten once that work has run — order the next H2D copy after it
                    copy_stream.wait_stream(torch.cuda.current_stream(self.device))
                    ring[prev][6].set()
                prev = si
                yield self._views(ring[si][4], ring[si][5], n)
        finally:
            stop.set()
            if prev is not None:
                ring[prev][6].set()
