"""The one-launch kernels of the row-sharded step (round 4) on a real MI355X against the launches they
replace and against the CPU restatements of tests/_cpu_emul.py: index work bit-exact, row moves
bit-exact, gradient sums bit-exact with the round-2 reduction (same ascending order), squared-norm
partials to fp32 summation tolerance."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import _cpu_emul as E  # noqa: E402
from fuxictr_amd import _lib, ops  # noqa: E402

DEV = "cuda:0"


def _dev(x, dtype=None):
    t = torch.as_tensor(x)
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV).contiguous()


def _received_runs(rng, R, L, rps):
    """What an owner receives: R ascending runs of local rows, duplicates across runs, pad rows (= rps)
    at the tails; one empty and one full run when there is room."""
    runs = []
    for r in range(R):
        n_valid = [0, L][r] if r < 2 and L > 1 else int(rng.integers(0, L + 1))
        n_valid = min(n_valid, rps)
        vals = np.sort(rng.choice(rps, size=n_valid, replace=False)) if n_valid else np.zeros(0, np.int64)
        runs.append(np.concatenate([vals, np.full(L - n_valid, rps)]))
    return np.concatenate(runs).astype(np.int32)


@pytest.mark.parametrize("series", [False, True])
@pytest.mark.parametrize("R,L,rps,dims,catchup", [
    (8, 2496, 50000, (16, 1), True), (1, 4000, 9000, (16, 1), True), (2, 300, 100, (16, 1), False),
    (3, 64, 5000, (8,), True), (4, 500, 777, (10, 1), True), (2, 128, 64, (16,), False)])
def test_owner_fetch_rows_equals_catchup_plus_gather(R, L, rps, dims, catchup, series):
    """fx_owner_fetch_rows == fx_adam_catchup_rows followed by one gather per table group (the round-2
    sequence): tables, moments, row stamps and the send block BIT FOR BIT; pad entries and pad columns
    of the block are zero; the extra zero row is cleared.  (Round 5 had relaxed this to 2e-6 for the
    D = 16 (+ D = 1) tables: fx_adam_catchup_rows ran the quad replay, the owner fetch the plain one.  Round 6:
    the owner fetch calls fx_catchup_quad itself, and both sum gaps beyond FX_SERIES_KDIR from the series
    table with per-element arithmetic that does not depend on the lane layout — a row is caught up to the
    same bits on 1 rank and on N.)  series: the scalar block carries the Adam series table and the rows are
    9 ... 700 steps behind, so both the table path and the short replays run."""
    rng = np.random.default_rng(R * 7919 + L)
    g = torch.Generator().manual_seed(R + L)
    idx = _received_runs(rng, R, L, rps)
    n = R * L
    ws = torch.empty(ops.dedup_workspace_bytes(n), dtype=torch.uint8, device=DEV)
    odd = ops.dedup_sorted_runs(_dev(idx).view(n, 1), R, rps + 1, rps, ws)
    scal = ops.new_scalars(DEV, lr=1e-2, series=series)
    n_open = 700 if series else 9
    scal.view(torch.int32)[_lib.SC_STEP] = n_open - 1
    ops.opt_begin_step(scal)                                       # step = 9 (700 with the series table)
    if series:
        assert int(scal.view(torch.int32)[_lib.SC_SERIES_TCAP].item()) > 0
    width = sum(dims) if len(dims) == 1 else -(-sum(dims) // 4) * 4
    offs, o = [], 0
    for D in dims:
        offs.append(o)
        o += D

    def make():
        sts = []
        gg = torch.Generator().manual_seed(1234)
        for D in dims:
            t = torch.randn(rps + 1, D, generator=gg)
            t[rps] = 0
            m = torch.randn(rps + 1, D, generator=gg) * 1e-2
            v = torch.rand(rps + 1, D, generator=gg) * 1e-3
            last = torch.randint(0, n_open, (rps + 1,), generator=gg).int()
            sts.append(ops.RowState(_dev(t), _dev(m), _dev(v), _dev(last), D))
        return sts
    a, b = make(), make()
    send = torch.full((n, width), 7.0, device=DEV)
    zero_row = torch.full((width,), 3.0, device=DEV)
    ops.owner_fetch_rows(a, offs, odd, send, catchup, scal, zero_row=zero_row)
    # reference sequence on the copies
    if catchup:
        ops.adam_catchup_rows(b, odd, -1, scal)
    ref = torch.zeros(n, width, device=DEV)
    ii = torch.from_numpy(idx.astype(np.int64)).to(DEV)
    for st, off in zip(b, offs):
        ref[:, off:off + st.D] = st.table[ii]                      # pad row of the table is zero
    torch.cuda.synchronize()
    def same(x, y):
        return torch.equal(x, y)
    assert same(send, ref)
    assert float(zero_row.abs().max()) == 0.0
    for x, y in zip(a, b):
        assert same(x.table, y.table) and same(x.m, y.m) and same(x.v, y.v)
        assert torch.equal(x.last_step, y.last_step)
    _ = g


@pytest.mark.parametrize("N,cap,n_max,dims", [(8, 2560, 20000, (16, 1)), (1, 4096, 3000, (16, 1)),
                                               (2, 64, 100, (8,)), (3, 128, 300, (8, 4, 1))])
def test_fill_grad_block_and_owner_grad_reduce(N, cap, n_max, dims):
    """fx_fill_grad_block: every slot holds its unique key's gradient rows (zeros for empty slots, for
    a group without gradient, for pad columns) == the zero fill + per-group scatter it replaces.
    fx_owner_grad_reduce: G per owned row == fx_emb_grad_reduce per group (bit-exact: same ascending
    order), squared-norm partials add up to sum G^2."""
    rng = np.random.default_rng(N * 100 + cap)
    n_slots = N * cap
    nu = min(n_max, n_slots - 7)
    slots = rng.choice(n_slots, size=nu, replace=False)
    slot_uniq = np.full(n_slots, -1, np.int32)
    slot_uniq[slots] = np.arange(nu, dtype=np.int32)
    width = sum(dims) if len(dims) == 1 else -(-sum(dims) // 4) * 4
    offs, o = [], 0
    for D in dims:
        offs.append(o)
        o += D
    g = torch.Generator().manual_seed(cap)
    Gs = [torch.randn(n_max, D, generator=g) for D in dims]
    tables = [(_dev(G), D, off) for G, D, off in zip(Gs, dims, offs)]
    if len(dims) == 3:
        tables[1] = (None, dims[1], offs[1])                       # a group without gradient this step
    block = torch.full((n_slots, width), 9.0, device=DEV)
    ops.fill_grad_block(tables, _dev(slot_uniq), block)
    ref = torch.zeros(n_slots, width)
    E.fill_grad_block([(None if t is None else t.cpu(), D, off) for t, D, off in tables],
                      torch.from_numpy(slot_uniq), ref)
    torch.cuda.synchronize()
    assert torch.equal(block.cpu(), ref)

    # owner side: the block as R = N received runs of local rows
    rps = 4 * cap
    idx = _received_runs(rng, N, cap, rps)
    ws = torch.empty(ops.dedup_workspace_bytes(n_slots), dtype=torch.uint8, device=DEV)
    odd = ops.dedup_sorted_runs(_dev(idx).view(n_slots, 1), N, rps + 1, rps, ws)
    grecv = torch.randn(n_slots, width, generator=g).to(DEV)
    outs = [(None if t is None else torch.full((odd.n_max, D), 5.0, device=DEV), D, off)
            for (t, D, off) in tables]
    sq = torch.full((ops.owner_grad_reduce_partials(odd.n_max),), 2.0, device=DEV)
    ops.owner_grad_reduce(grecv, odd, outs, sq)
    nuo = int(odd.n_unique.item())
    total = 0.0
    zero_off = _dev([0], torch.int64)
    for G_own, D, off in outs:
        if G_own is None:
            continue
        G_ref = torch.empty(odd.n_max, D, device=DEV)
        sq_ref = torch.empty(ops.emb_grad_reduce_partials(odd.n_max, D), device=DEV)
        scratch = torch.zeros(ops.emb_grad_reduce_scratch_ints(odd.n_max), dtype=torch.int32, device=DEV)
        ops.emb_grad_reduce(grecv[:, off:off + D], grecv.stride(0), zero_off, 1, D, odd, G_ref, sq_ref,
                            scratch)
        assert torch.equal(G_own[:nuo], G_ref[:nuo]), D
        total += float((G_own[:nuo].double() ** 2).sum())
    got = float(sq.double().sum())
    assert abs(got - total) <= 1e-5 * max(1.0, total), (got, total)


@pytest.mark.parametrize("want_lr,want_fm", [(True, True), (False, False), (True, False)])
def test_emb_fm_fwd_reads_a_strided_row_block_in_place(want_lr, want_fm):
    """fx_emb_fm_fwd with table / table1 = column ranges of one wider block (the received rows of a
    row-sharded exchange, row stride 20 floats) == the same call on contiguous copies, bit for bit."""
    g = torch.Generator().manual_seed(5)
    B, C, Fd, D = 1000, 7, 3, 16
    n_rows = 3001
    block = torch.randn(n_rows, 20, generator=g).to(DEV)
    block[n_rows - 1] = 0
    tab, tab1 = block[:, 0:16], block[:, 16:17]
    ids = torch.randint(0, n_rows, (B, C), generator=g).int().to(DEV)
    base = torch.zeros(C, dtype=torch.int64, device=DEV)
    vocab = torch.full((C,), n_rows, dtype=torch.int32, device=DEV)
    out_off = _dev([(c + Fd) * D for c in range(C)], torch.int64)
    num_off = _dev([j * D for j in range(Fd)], torch.int64)
    dense = torch.rand(B, Fd, generator=g).to(DEV)
    num_w = torch.randn(Fd, D, generator=g).to(DEV)
    num_w1 = torch.randn(Fd, 1, generator=g).to(DEV)
    bias = torch.randn(1, generator=g).to(DEV)
    scal = ops.new_scalars(DEV)

    def run(t, t1):
        out = torch.empty(B, (C + Fd) * D, device=DEV)
        lr = torch.empty(B, 1, device=DEV) if want_lr else None
        fm = torch.empty(B, 1, device=DEV) if want_fm else None
        fl = torch.empty(B, 1, device=DEV) if (want_lr and want_fm) else None
        S = torch.empty(B, D, device=DEV) if want_fm else None
        ops.emb_fm_fwd(t, D, ids, base, vocab, out_off, dense, num_w, num_off, out, scal,
                       table1=t1 if want_lr else None, num_w1=num_w1 if want_lr else None,
                       bias1=bias if want_lr else None, lr_out=lr, fm_out=fm, fm_lr_out=fl, S=S)
        return [x for x in (out, lr, fm, fl, S) if x is not None]
    a = run(tab, tab1)
    b = run(tab.contiguous(), tab1.contiguous())
    torch.cuda.synchronize()
    for x, y in zip(a, b):
        assert torch.equal(x, y)


@pytest.mark.parametrize("N,B", [(2, 4096), (8, 4096), (3, 777), (1, 512)])
def test_shard_plan_slot_uniq_is_the_inverse_of_uniq_slot(N, B):
    rng = np.random.default_rng(N * B + 1)
    vocab = [50, 3, 7000, 911, 12]
    C = len(vocab)
    ids = np.stack([rng.integers(0, v, B) for v in vocab], 1).astype(np.int32)
    base = np.concatenate([[0], np.cumsum(vocab)[:-1]]).astype(np.int64)
    total = int(sum(vocab))
    n = B * C
    ws = torch.empty(ops.dedup_workspace_bytes(n), dtype=torch.uint8, device=DEV)
    dd = ops.dedup(_dev(ids), _dev(base), _dev(np.array(vocab, np.int32)), _dev(np.zeros(C, np.int32)),
                   total, ws, want_uid=True, columns_sorted=True)
    cap = int(np.ceil(1.5 * n / N / 64) * 64 + 64)
    send_idx = torch.empty(N * cap, dtype=torch.int32, device=DEV)
    uniq_slot = torch.empty(n, dtype=torch.int32, device=DEV)
    slot_uniq = torch.full((N * cap,), 12345, dtype=torch.int32, device=DEV)
    lookup_slot = torch.empty(B, C, dtype=torch.int32, device=DEV)
    scal = ops.new_scalars(DEV)
    pws = torch.empty(ops.shard_plan_workspace_ints(n, N), dtype=torch.int32, device=DEV)
    ops.shard_plan(dd, N, total, cap, send_idx, uniq_slot, lookup_slot, scal, global_keys=True,
                   workspace=pws, slot_uniq=slot_uniq)
    torch.cuda.synchronize()
    assert int(scal.view(torch.int32)[_lib.SC_ERR]) & _lib.FX_FLAG_A2A_OVERFLOW == 0
    nu = int(dd.n_unique.item())
    us, su = uniq_slot.cpu().numpy(), slot_uniq.cpu().numpy()
    assert (us[nu:] == N * cap).all()
    np.testing.assert_array_equal(su[us[:nu]], np.arange(nu))
    assert int((su >= 0).sum()) == nu and (su[su < 0] == -1).all()
    # and the CPU restatement agrees on every output
    s2 = torch.empty(N * cap, dtype=torch.int32)
    u2 = torch.empty(n, dtype=torch.int32)
    su2 = torch.empty(N * cap, dtype=torch.int32)
    l2 = torch.empty(B, C, dtype=torch.int32)
    dd_c = type("D", (), {})()
    for k in ("uniq_row", "n_unique", "sorted_pos", "sorted_uid"):
        setattr(dd_c, k, getattr(dd, k).cpu())
    # the fast path marks padding lookups by pos = 0xFFFFFFFF (-1 as int32): the restatement scatters by
    # position, so park those entries on a lookup that IS padding and compare the valid lookups
    valid = ids != 0
    pos = dd_c.sorted_pos.long()
    keep = pos >= 0
    if not bool(keep.all()):
        parked = int(np.flatnonzero(~valid.reshape(-1))[0])
        dd_c.sorted_pos = torch.where(keep, pos, torch.full_like(pos, parked)).int()
        dd_c.sorted_uid = torch.where(keep, dd_c.sorted_uid.long(), torch.full_like(pos, -1)).int()
    E.shard_plan(dd_c, N, total, cap, s2, u2, l2, E.new_scalars("cpu"), global_keys=True, slot_uniq=su2)
    np.testing.assert_array_equal(send_idx.cpu().numpy(), s2.numpy())
    np.testing.assert_array_equal(us[:nu], u2.numpy()[:nu])
    np.testing.assert_array_equal(su, su2.numpy())
    np.testing.assert_array_equal(lookup_slot.cpu().numpy()[valid], l2.numpy()[valid])


@pytest.mark.parametrize("catchup", [True, False])
def test_owner_fetch_rows_of_a_bf16_table(catchup):
    """Round 6: the owner fetch takes bf16 tables (`emb_dtype: bf16` + `shard: row`).  What travels is the row as
    an unsharded bf16 table would hold it after the same catch-up — rounded to nearest-even bf16, then widened —
    so: send block == gather of the tables that fx_adam_catchup_rows left behind, bit for bit; the D = 1 table of
    the first-order term stays fp32."""
    R, L, rps = 4, 600, 3000
    rng = np.random.default_rng(77)
    idx = _received_runs(rng, R, L, rps)
    n = R * L
    ws = torch.empty(ops.dedup_workspace_bytes(n), dtype=torch.uint8, device=DEV)
    odd = ops.dedup_sorted_runs(_dev(idx).view(n, 1), R, rps + 1, rps, ws)
    scal = ops.new_scalars(DEV, lr=1e-2, series=True)
    scal.view(torch.int32)[_lib.SC_STEP] = 399
    ops.opt_begin_step(scal)

    def make():
        gg = torch.Generator().manual_seed(4321)
        sts = []
        for D, dt in ((16, torch.bfloat16), (1, torch.float32)):
            t = torch.randn(rps + 1, D, generator=gg)
            t[rps] = 0
            m = torch.randn(rps + 1, D, generator=gg) * 1e-2
            v = torch.rand(rps + 1, D, generator=gg) * 1e-3
            last = torch.randint(0, 400, (rps + 1,), generator=gg).int()
            sts.append(ops.RowState(_dev(t).to(dt), _dev(m), _dev(v), _dev(last), D))
        return sts
    a, b = make(), make()
    send = torch.full((n, 20), 7.0, device=DEV)
    zero_row = torch.full((20,), 3.0, device=DEV)
    ops.owner_fetch_rows(a, [0, 16], odd, send, catchup, scal, zero_row=zero_row)
    if catchup:
        ops.adam_catchup_rows(b, odd, -1, scal)
    ref = torch.zeros(n, 20, device=DEV)
    ii = torch.from_numpy(idx.astype(np.int64)).to(DEV)
    ref[:, 0:16] = b[0].table[ii].float()
    ref[:, 16:17] = b[1].table[ii]
    torch.cuda.synchronize()
    assert torch.equal(send, ref)
    assert float(zero_row.abs().max()) == 0.0
    for x, y in zip(a, b):
        assert torch.equal(x.table, y.table) and torch.equal(x.m, y.m) and torch.equal(x.v, y.v)
        assert torch.equal(x.last_step, y.last_step)
    if catchup:
        assert not torch.equal(a[0].table, make()[0].table)          # rows did move
