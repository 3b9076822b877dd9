"""Where does k_catchup_rows' time go?  The product kernel (fx_adam_catchup_rows) on synthetic row states,
one variable at a time: N unique rows of an R-row D = 16 table in the row-record layout, every launch on a fresh
row set, launches serialised in one hipGraph (what the training step does), per-launch time from HIP events.
    variants   uptodate  last == upto            loads only (the quad returns before any arithmetic)
               never     m == 0                   loads + stamp store (rows that never had a gradient)
               k1        1 missed step            the short path (k <= 12: summed step by step)
               k12       12 missed steps          the longest short path
               k100      100 missed steps         the series table, one segment per entry (t >= 128)
               early     100 missed, last < 128   the series table's early entries (up to 8 segments)
               mix       gaps as in an aged run   1 .. 300, power-law
usage: python scripts/catchup_probe.py [R] [N]"""
import sys

import numpy as np
import torch

from fuxictr_amd import _lib, ops
from fuxictr_amd.layers import _TableGroup

DEV = "cuda:0"
R = int(sys.argv[1]) if len(sys.argv) > 1 else 33762603
N = int(sys.argv[2]) if len(sys.argv) > 2 else 25000
D, SETS, STEP = 16, 16, 320
W = _TableGroup.record_width(D)


class DD(object):
    pass


def main():
    g = torch.Generator(device=DEV).manual_seed(1)
    rec = torch.zeros(R, W, device=DEV)
    rec[:, :D].normal_(generator=g)
    table, m, v = rec[:, :D], rec[:, D:2 * D], rec[:, 2 * D:3 * D]
    last = rec.view(torch.int32)[:, 3 * D]
    scal = ops.new_scalars(DEV, series=True)
    scal.view(torch.int32)[_lib.SC_STEP] = STEP
    rng = np.random.default_rng(0)
    rows = [torch.from_numpy(np.sort(rng.choice(R, N, replace=False)).astype(np.int64)).to(DEV) for _ in range(SETS)]
    dds = []
    for r in rows:
        dd = DD()
        dd.uniq_row = r.to(torch.int32)           # (bit pattern of the uint32 rows)
        dd.n_unique = torch.tensor([N], dtype=torch.int32, device=DEV)
        dd.n_max = 106496 if N <= 106496 else N
        dds.append(dd)
    st = [ops.RowState(table, m, v, last, D)]
    upto = STEP - 1
    out = []
    for name in ("uptodate", "never", "k1", "k12", "k100", "early", "mix"):
        def prepare():
            for r in rows:
                if name == "never":
                    m[r] = 0.0
                    v[r] = 0.0
                else:
                    m[r] = torch.randn(N, D, device=DEV, generator=g) * 1e-3
                    v[r] = torch.rand(N, D, device=DEV, generator=g) * 1e-6 + 1e-10
                if name == "uptodate":
                    last[r] = upto
                elif name in ("never", "k100"):
                    last[r] = upto - 100
                elif name == "k1":
                    last[r] = upto - 1
                elif name == "k12":
                    last[r] = upto - 12
                elif name == "early":
                    last[r] = 20
                else:
                    gaps = np.minimum((300 * rng.random(N) ** 3).astype(np.int64) + 1, 300)
                    last[r] = torch.from_numpy(upto - gaps).to(DEV).to(torch.int32)
        prepare()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            ops.adam_catchup_rows(st, dds[0], -1, scal)      # warm
            side.synchronize()
            prepare()
            side.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                for dd in dds:
                    ops.adam_catchup_rows(st, dd, -1, scal)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        graph.replay()            # the first replay does the variant's work; later ones find the rows up to date
        e1.record()
        torch.cuda.synchronize()
        t_first = e0.elapsed_time(e1) * 1e3 / SETS
        e0.record()
        graph.replay()
        e1.record()
        torch.cuda.synchronize()
        t_again = e0.elapsed_time(e1) * 1e3 / SETS
        out.append((name, t_first, t_again))
        print("%-9s %7.2f us per launch (rows now up to date: %6.2f us)   R = %d, N = %d" % (name, t_first, t_again, R, N),
              flush=True)
    return out


if __name__ == "__main__":
    main()
