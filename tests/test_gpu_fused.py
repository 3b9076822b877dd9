"""The fused sparse front / back end (csrc/fx_fused.hip) on a real MI355X, entry point by entry point,
against the unfused entry points they replace (themselves checked against the oracle in
tests/test_gpu_kernels.py) and against torch-CPU restatements.  Index work and copies are bit-exact;
fp32 sums carry the tolerance written next to each check.  Edge cases the domain has: padding ids,
bad ids, a column whose every lookup hits the same row (one run of B lookups, spread over all lane
groups of a workgroup), runs that straddle workgroup pieces, ragged B, D = 1 / 8 / 10 / 16 / 40."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from fuxictr_amd import _lib, ops  # noqa: E402

DEV = "cuda:0"


def _dev(x, dtype=None):
    t = torch.as_tensor(x)
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV).contiguous()


def _schema(vocabs):
    bases = np.concatenate([[0], np.cumsum(vocabs)[:-1]]).astype(np.int64)
    return bases, int(sum(vocabs))


def _ids(rng, B, vocabs, mode):
    cols = []
    for c, v in enumerate(vocabs):
        if mode == "same" and c == 1:
            col = np.full(B, min(2, v - 1))                  # one run of B lookups
        elif mode == "power":
            col = np.minimum((v * rng.random(B) ** 3).astype(np.int64) + 1, v - 1)
        else:
            col = rng.integers(0, v, B)
        cols.append(col)
    ids = np.stack(cols, axis=1).astype(np.int64)
    if mode == "pad":
        ids[:] = 0                                           # every lookup is padding_idx
    return ids


def _tables(rng, R, D):
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    return (torch.randn(R, D, generator=g), torch.randn(R, D, generator=g) * 0.1,
            torch.rand(R, D, generator=g) * 0.01)


@pytest.mark.parametrize("mode", ["uniform", "power", "same", "pad"])
@pytest.mark.parametrize("B", [1, 777, 4096, 8192])
def test_dedup_catchup_equals_dedup_plus_catchup(B, mode):
    rng = np.random.default_rng(B + len(mode))
    vocabs = [5, 40, 3000, 9, 70000]
    pads = [0, 0, 0, -1, 0]
    bases, R = _schema(vocabs)
    ids = _ids(rng, B, vocabs, mode)
    if mode == "uniform" and B > 10:
        ids[3, 2] = vocabs[2] + 5                            # a bad id (reads as "contributes nothing")
    C = len(vocabs)
    ws = torch.empty(ops.dedup_workspace_bytes(B * C), dtype=torch.uint8, device=DEV)
    args = (_dev(ids, torch.int32), _dev(bases, torch.int64), _dev(vocabs, torch.int32),
            _dev(pads, torch.int32))
    # state of two table groups (D = 16 and D = 1) with rows last updated at various steps
    st = {}
    for D in (16, 1):
        t, m, v = _tables(rng, R, D)
        last = torch.from_numpy(rng.integers(0, 7, R).astype(np.int32))
        st[D] = [x.clone() for x in (t, m, v)] + [last.clone()]
    scal_a = ops.new_scalars(DEV)
    scal_b = ops.new_scalars(DEV)
    for s in (scal_a, scal_b):
        s.view(torch.int32)[_lib.SC_STEP] = 8
    # reference path: begin-step, generic de-dup (column fast path), one catch-up launch per group
    ops.opt_begin_step(scal_a)
    dd_a = ops.dedup(*args, R, ws, columns_sorted=True, want_uid=True)
    ref = {}
    for D in (16, 1):
        t, m, v, last = (x.clone().to(DEV) for x in st[D])
        ops.adam_catchup(t, m, v, last, D, dd_a, R, -1, scal_a)
        ref[D] = (t, m, v, last)
    # fused path
    got = {D: tuple(x.clone().to(DEV) for x in st[D]) for D in (16, 1)}
    dd_b = ops.dedup_catchup(args[0], args[1], args[2], args[3], ws,
                             [ops.RowState(*got[D], D) for D in (16, 1)], scal_b,
                             begin_scal=scal_b, want_uid=True)
    torch.cuda.synchronize()
    assert torch.equal(scal_a.cpu(), scal_b.cpu())           # the fused begin-step
    nu = int(dd_a.n_unique.item())
    assert int(dd_b.n_unique.item()) == nu
    for name in ("sorted_key", "sorted_pos", "sorted_uid"):
        assert torch.equal(getattr(dd_a, name), getattr(dd_b, name)), name
    assert torch.equal(dd_a.uniq_row[:nu], dd_b.uniq_row[:nu])
    assert torch.equal(dd_a.seg_start[:nu + 1], dd_b.seg_start[:nu + 1])
    for D in (16, 1):
        for a, b, what in zip(ref[D], got[D], ("table", "m", "v", "last_step")):
            if what == "last_step":
                assert torch.equal(a, b), (D, what)
            else:
                # (round 5: the D = 16 + D = 1 pair leaves through the quad replay — fx_catchup_quad: the same
                # terms summed before they meet p, the moments' decay as one power — not the two plain replays
                # the single-table entry point runs: equal to fp32 rounding, not to the bit)
                assert (a - b).abs().max().item() <= 2e-6 * max(1.0, a.abs().max().item()), (D, what)


@pytest.mark.parametrize("D", [16, 8, 10, 1, 40])
@pytest.mark.parametrize("with_lr,with_fm", [(True, True), (True, False), (False, True), (False, False)])
def test_emb_fm_fwd_equals_the_three_unfused_kernels(D, with_lr, with_fm):
    rng = np.random.default_rng(D * 4 + with_lr * 2 + with_fm)
    g = torch.Generator().manual_seed(D)
    vocabs = [50, 3, 1000, 7, 20011]
    bases, R = _schema(vocabs)
    B, C, Fd = 1003, len(vocabs), 3
    table, num_w = torch.randn(R, D, generator=g), torch.randn(Fd, D, generator=g)
    table1, num_w1 = torch.randn(R, 1, generator=g), torch.randn(Fd, 1, generator=g)
    bias1 = torch.randn(1, generator=g)
    ids = _ids(rng, B, vocabs, "power")
    dense = torch.rand(B, Fd, generator=g)
    slots_c, slots_n = [1, 2, 4, 5, 7], [0, 3, 6]
    F = 8
    common = (_dev(ids, torch.int32), _dev(bases, torch.int64), _dev(vocabs, torch.int32),
              _dev([s * D for s in slots_c], torch.int64), _dev(dense))
    scal = ops.new_scalars(DEV)
    # unfused
    rec_a = torch.empty(B, F * D, device=DEV)
    ops.emb_gather_fwd(_dev(table), D, *common, _dev(num_w),
                       _dev([s * D for s in slots_n], torch.int64), rec_a, scal)
    lr_a = torch.empty(B, 1, device=DEV)
    ops.lr_fwd(_dev(table1), common[0], common[1], common[2], common[4], _dev(num_w1), _dev(bias1),
               lr_a, scal)
    fm_a = torch.empty(B, 1, device=DEV)
    ops.fm_fwd(rec_a, F, D, None, fm_a)
    # fused
    rec_b = torch.full((B, F * D), 5.0, device=DEV)
    lr_b = torch.empty(B, 1, device=DEV) if with_lr else None
    fm_b = torch.empty(B, 1, device=DEV) if with_fm else None
    fml_b = torch.empty(B, 1, device=DEV) if (with_lr and with_fm) else None
    S = torch.empty(B, D, device=DEV) if with_fm else None
    ops.emb_fm_fwd(_dev(table), D, common[0], common[1], common[2], common[3], common[4],
                   _dev(num_w), _dev([s * D for s in slots_n], torch.int64), rec_b, scal,
                   table1=_dev(table1) if with_lr else None, num_w1=_dev(num_w1) if with_lr else None,
                   bias1=_dev(bias1) if with_lr else None, lr_out=lr_b, fm_out=fm_b, fm_lr_out=fml_b,
                   S=S)
    torch.cuda.synchronize()
    assert torch.equal(rec_a, rec_b)                         # the record is a copy: bit-exact
    e64 = rec_a.cpu().double().view(B, F, D)
    if with_lr:
        # fp32 sums of 8 terms in a different order: |d| <= 8 eps * sum|terms|
        np.testing.assert_allclose(lr_b.cpu().numpy(), lr_a.cpu().numpy(), rtol=0, atol=2e-5)
    if with_fm:
        ref = 0.5 * ((e64.sum(1) ** 2) - (e64 ** 2).sum(1)).sum(-1, keepdim=True)
        scale = float((e64 ** 2).sum((1, 2)).max())
        assert float((fm_b.cpu().double() - ref).abs().max()) <= 3e-6 * scale
        assert float((fm_a.cpu().double() - ref).abs().max()) <= 3e-6 * scale
        np.testing.assert_allclose(S.cpu().double().numpy(), e64.sum(1).numpy(), rtol=0, atol=2e-5)
    if with_lr and with_fm:
        np.testing.assert_allclose(fml_b.cpu().numpy(), (fm_b + lr_b).cpu().numpy(), rtol=0, atol=0)
    assert int(scal.view(torch.int32)[_lib.SC_ERR]) == 0


@pytest.mark.parametrize("C,Fd,D,B", [(26, 13, 16, 4096), (26, 13, 16, 32769), (64, 0, 16, 2049),
                                      (5, 3, 8, 1000), (5, 3, 10, 513), (7, 2, 1, 300), (6, 2, 40, 257),
                                      (0, 9, 16, 100), (70, 3, 16, 500), (26, 13, 16, 1)])
@pytest.mark.parametrize("with_lr,with_fm", [(True, True), (False, False)])
def test_emb_fm_fwd_round4_kernel_is_bit_identical_to_the_first_version(C, Fd, D, B, with_lr, with_fm):
    """k_emb_fm_fwd2 (per-wave constants, ids by shuffle, every row load of a sample in flight, two
    samples per wave at large B) == k_emb_fm_fwd (FX_EMB_FWD2=0), every output bit for bit: the record,
    the first-order term, the FM term, their sum, the field sums; bad / padding ids included; shapes
    outside the new kernel's range (C > 64) take the first version either way."""
    import os
    rng = np.random.default_rng(C * 1000 + D + B)
    g = torch.Generator().manual_seed(C + D + B)
    vocabs = [int(v) for v in rng.integers(3, 5000, C)]
    bases, R = _schema(vocabs) if C else (np.zeros(0, np.int64), 1)
    table, num_w = torch.randn(max(R, 1), D, generator=g), torch.randn(max(Fd, 1), D, generator=g)
    table1, num_w1 = torch.randn(max(R, 1), 1, generator=g), torch.randn(max(Fd, 1), 1, generator=g)
    bias1 = torch.randn(1, generator=g)
    ids = _ids(rng, B, vocabs, "power") if C else None
    if C and B > 10:
        ids[3, 0] = vocabs[0] + 7                                # a bad id: zeros + the error flag
        ids[5, C - 1] = -1
    dense = torch.rand(B, max(Fd, 1), generator=g)[:, :Fd].contiguous() if Fd else None
    perm = rng.permutation(C + Fd)
    slots_c, slots_n = perm[:C], perm[C:]
    F = C + Fd
    outs = []
    for flag in ("0", "1"):
        os.environ["FX_EMB_FWD2"] = flag
        scal = ops.new_scalars(DEV)
        rec = torch.full((B, F * D), 5.0, device=DEV)
        lr = torch.empty(B, 1, device=DEV) if with_lr else None
        fm = torch.empty(B, 1, device=DEV) if with_fm else None
        fml = torch.empty(B, 1, device=DEV) if (with_lr and with_fm) else None
        S = torch.empty(B, D, device=DEV) if with_fm else None
        ops.emb_fm_fwd(_dev(table) if C else None, D, _dev(ids, torch.int32) if C else None,
                       _dev(bases, torch.int64) if C else None, _dev(vocabs, torch.int32) if C else None,
                       _dev([s * D for s in slots_c], torch.int64) if C else None,
                       _dev(dense) if Fd else None, _dev(num_w[:Fd]) if Fd else None,
                       _dev([s * D for s in slots_n], torch.int64) if Fd else None, rec, scal,
                       table1=_dev(table1) if (with_lr and C) else None,
                       num_w1=_dev(num_w1[:Fd]) if (with_lr and Fd) else None,
                       bias1=_dev(bias1) if with_lr else None, lr_out=lr, fm_out=fm, fm_lr_out=fml, S=S)
        torch.cuda.synchronize()
        outs.append([rec, lr, fm, fml, S, int(scal.view(torch.int32)[_lib.SC_ERR])])
    os.environ.pop("FX_EMB_FWD2")
    for x, y in zip(outs[0][:5], outs[1][:5]):
        assert (x is None) == (y is None)
        if x is not None:
            assert torch.equal(x, y)
    assert outs[0][5] == outs[1][5]
    if C and B > 10:
        assert outs[1][5] & _lib.FX_FLAG_BAD_ID


@pytest.mark.parametrize("fwd2", ["1", "0"])
def test_emb_fm_fwd_clears_the_reserved_slots(fwd2):
    """Reserved slots of the record (DIN's hole, DLRM's tail: filled by a later kernel of the step) are
    cleared by the gather launch, by both forms of the kernel; every other float of the row is what the
    launch without reserved ranges writes."""
    import os
    rng = np.random.default_rng(11)
    g = torch.Generator().manual_seed(11)
    vocabs = [50, 3, 1000, 7]
    bases, R = _schema(vocabs)
    B, C, Fd, D = 513, 4, 2, 16
    table, num_w = torch.randn(R, D, generator=g), torch.randn(Fd, D, generator=g)
    ids = _ids(rng, B, vocabs, "power")
    dense = torch.rand(B, Fd, generator=g)
    # slots: [n0, c0, c1, HOLE, c2, c3, n1, TAIL, TAIL]
    slots_c, slots_n, n_slots = [1, 2, 4, 5], [0, 6], 9
    os.environ["FX_EMB_FWD2"] = fwd2
    outs = []
    for ranges in ((), ((3 * D, D), (7 * D, 2 * D))):
        scal = ops.new_scalars(DEV)
        rec = torch.full((B, n_slots * D), 5.0, device=DEV)
        ops.emb_fm_fwd(_dev(table), D, _dev(ids, torch.int32), _dev(bases, torch.int64),
                       _dev(vocabs, torch.int32), _dev([s_ * D for s_ in slots_c], torch.int64), _dev(dense),
                       _dev(num_w), _dev([s_ * D for s_ in slots_n], torch.int64), rec, scal,
                       zero_ranges=ranges)
        torch.cuda.synchronize()
        outs.append(rec.view(B, n_slots, D).cpu())
    os.environ.pop("FX_EMB_FWD2")
    plain, cleared = outs
    keep = [0, 1, 2, 4, 5, 6]
    assert torch.equal(plain[:, keep], cleared[:, keep])
    assert float(plain[:, [3, 7, 8]].min()) == 5.0                 # untouched without ranges
    assert float(cleared[:, [3, 7, 8]].abs().max()) == 0.0


def _bwd_reference(drec, rec, S, g_fm, g_lr, ids, bases, pads, vocabs, slots_c, slots_n, dense, D):
    """float64 restatement: the dense [R, D] gradient autograd would build, then the unique rows."""
    B, C = ids.shape
    full = torch.zeros_like(rec, dtype=torch.float64) if drec is None else drec.double().clone()
    if g_fm is not None:
        e = rec.double().view(B, -1, D)
        full = full + (g_fm.double().view(B, 1, 1) * (S.double().view(B, 1, D) - e)).reshape(B, -1)
    R = int(sum(vocabs))
    Gd = torch.zeros(R, D, dtype=torch.float64)
    G1d = torch.zeros(R, dtype=torch.float64)
    for c in range(C):
        ok = (ids[:, c] >= 0) & (ids[:, c] < vocabs[c]) & (ids[:, c] != pads[c])
        rows = torch.from_numpy(ids[ok, c] + bases[c])
        Gd.index_add_(0, rows, full[torch.from_numpy(ok), slots_c[c] * D:(slots_c[c] + 1) * D])
        if g_lr is not None:
            G1d.index_add_(0, rows, g_lr.double().view(-1)[torch.from_numpy(ok)])
    dnum = torch.stack([(dense[:, j:j + 1].double() * full[:, s * D:(s + 1) * D]).sum(0)
                        for j, s in enumerate(slots_n)])
    return Gd, G1d, dnum


@pytest.mark.parametrize("mode", ["power", "same", "pad", "uniform"])
@pytest.mark.parametrize("D,B", [(16, 4096), (16, 333), (8, 1000), (10, 513), (1, 700), (40, 257)])
@pytest.mark.parametrize("terms", ["drec+fm+lr", "drec", "fm+lr", "drec+lr"])
def test_emb_fm_bwd_equals_autograd(D, B, mode, terms):
    rng = np.random.default_rng(D * 7 + B + len(mode) + len(terms))
    g = torch.Generator().manual_seed(D + B)
    vocabs = [5, 40, 3000, 9, 70000]
    pads = [0, 0, 0, -1, 0]
    bases, R = _schema(vocabs)
    C, Fd = len(vocabs), 3
    slots_c, slots_n = [1, 2, 4, 5, 7], [0, 3, 6]
    F = 8
    ids = _ids(rng, B, vocabs, mode)
    dense = torch.rand(B, Fd, generator=g)
    rec = torch.randn(B, F * D, generator=g)
    S = rec.view(B, F, D).sum(1).contiguous()
    drec = torch.randn(B, F * D, generator=g) if "drec" in terms else None
    g_fm = torch.randn(B, 1, generator=g) if "fm" in terms else None
    g_lr = torch.randn(B, 1, generator=g) if "lr" in terms else None
    ws = torch.empty(ops.dedup_workspace_bytes(B * C), dtype=torch.uint8, device=DEV)
    scal = ops.new_scalars(DEV)
    dd = ops.dedup_catchup(_dev(ids, torch.int32), _dev(bases, torch.int64),
                           _dev(vocabs, torch.int32), _dev(pads, torch.int32), ws, [], scal,
                           want_uid=True)
    n_max = dd.n_max
    nparts = ops.emb_fm_bwd_partials(n_max, D)
    wsf = torch.empty(ops.emb_fm_bwd_workspace_floats(n_max, D, Fd), device=DEV)
    G = torch.full((n_max, D), 9.0, device=DEV)
    sq = torch.full((nparts,), 9.0, device=DEV)
    G1 = torch.full((n_max, 1), 9.0, device=DEV) if g_lr is not None else None
    sq1 = torch.full((nparts,), 9.0, device=DEV) if g_lr is not None else None
    dnum = torch.empty(Fd, D, device=DEV)
    dnum1 = torch.empty(Fd, 1, device=DEV) if g_lr is not None else None
    dbias = torch.empty(1, device=DEV) if g_lr is not None else None

    def d(t):
        return None if t is None else _dev(t)
    for rep in range(2):                                     # run-to-run determinism
        ops.emb_fm_bwd(d(drec), d(rec), d(S), d(g_fm), d(g_lr),
                       _dev([s * D for s in slots_c], torch.int64), C, D, dd, G, sq, G1, sq1,
                       _dev(dense), _dev([s * D for s in slots_n], torch.int64), B, dnum, dnum1, dbias,
                       wsf)
        torch.cuda.synchronize()
        snap = (G.clone(), sq.clone(), None if G1 is None else G1.clone())
        if rep:
            assert torch.equal(snap[0], first[0]) and torch.equal(snap[1], first[1])
        first = snap
    Gd, G1d, dnum_ref = _bwd_reference(drec, rec, S, g_fm, g_lr, ids, bases, pads, vocabs, slots_c,
                                       slots_n, dense, D)
    nu = int(dd.n_unique.item())
    rows = dd.uniq_row[:nu].cpu().long()
    # every term is O(1), a run has at most B of them: |d| <= run * eps * max|term| (fp32 sums)
    tol = 4e-7 * B * 3 + 2e-5
    got = torch.zeros(R, D, dtype=torch.float64)
    got[rows] = G[:nu].cpu().double()
    assert float((got - Gd).abs().max()) <= tol, float((got - Gd).abs().max())
    np.testing.assert_allclose(float(sq.cpu().double().sum()), float((Gd ** 2).sum()),
                               rtol=1e-4, atol=1e-6)
    if g_lr is not None:
        got1 = torch.zeros(R, dtype=torch.float64)
        got1[rows] = G1[:nu, 0].cpu().double()
        assert float((got1 - G1d).abs().max()) <= tol
        np.testing.assert_allclose(float(sq1.cpu().double().sum()), float((G1d ** 2).sum()),
                                   rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(dnum1.cpu().double().numpy().reshape(-1),
                                   (dense.double() * g_lr.double()).sum(0).numpy(), atol=tol)
        np.testing.assert_allclose(float(dbias.item()), float(g_lr.double().sum()), atol=tol)
    np.testing.assert_allclose(dnum.cpu().double().numpy(), dnum_ref.numpy(), atol=tol)


def test_emb_fm_bwd_equals_the_unfused_reduce_bitwise_on_short_runs():
    """Runs that lie inside one lane-group piece are summed in ascending lookup order — exactly
    what fx_emb_grad_reduce does — so with unique-ish ids both entry points give the same bits."""
    rng = np.random.default_rng(3)
    g = torch.Generator().manual_seed(3)
    D, B = 16, 2048
    vocabs = [500000, 400000, 300000]
    bases, R = _schema(vocabs)
    C = 3
    ids = _ids(rng, B, vocabs, "uniform")
    drec = torch.randn(B, C * D, generator=g)
    ws = torch.empty(ops.dedup_workspace_bytes(B * C), dtype=torch.uint8, device=DEV)
    scal = ops.new_scalars(DEV)
    dd = ops.dedup_catchup(_dev(ids, torch.int32), _dev(bases, torch.int64),
                           _dev(vocabs, torch.int32), _dev([0, 0, 0], torch.int32), ws, [], scal,
                           want_uid=True)
    off = _dev([0, D, 2 * D], torch.int64)
    Ga = torch.zeros(dd.n_max, D, device=DEV)
    sqa = torch.zeros(ops.emb_grad_reduce_partials(dd.n_max, D), device=DEV)
    Gb = torch.zeros(dd.n_max, D, device=DEV)
    sqb = torch.zeros(ops.emb_fm_bwd_partials(dd.n_max, D), device=DEV)
    wsf = torch.empty(ops.emb_fm_bwd_workspace_floats(dd.n_max, D, 0), device=DEV)
    scratch = torch.zeros(ops.emb_grad_reduce_scratch_ints(dd.n_max), dtype=torch.int32, device=DEV)
    ops.emb_grad_reduce(_dev(drec), C * D, off, C, D, dd, Ga, sqa, scratch)
    ops.emb_fm_bwd(_dev(drec), None, None, None, None, off, C, D, dd, Gb, sqb, None, None, None,
                   None, B, None, None, None, wsf)
    torch.cuda.synchronize()
    nu = int(dd.n_unique.item())
    assert torch.equal(Ga[:nu], Gb[:nu])


@pytest.mark.parametrize("dims", [(16, 1, 10), (16, 1)])
@pytest.mark.parametrize("kind", ["adam", "sgd"])
def test_sparse_update_multi_equals_one_launch_per_table(kind, dims):
    rng = np.random.default_rng(11)
    vocabs = [5, 40, 3000]
    bases, R = _schema(vocabs)
    B, C = 900, 3
    ids = _ids(rng, B, vocabs, "power")
    ws = torch.empty(ops.dedup_workspace_bytes(B * C), dtype=torch.uint8, device=DEV)
    scal = ops.new_scalars(DEV, lr=0.01)
    ops.opt_begin_step(scal)
    scal[_lib.SC_CLIP] = 0.7
    dd = ops.dedup_catchup(_dev(ids, torch.int32), _dev(bases, torch.int64),
                           _dev(vocabs, torch.int32), _dev([0, 0, 0], torch.int32), ws, [], scal)
    states_a, states_b = [], []
    for D in dims:
        t, m, v = _tables(rng, R, D)
        last = torch.zeros(R, dtype=torch.int32)
        G = torch.randn(dd.n_max, D)
        a = [x.clone().to(DEV) for x in (t, m, v, last)] + [G.to(DEV)]
        b = [x.clone().to(DEV) for x in (t, m, v, last)] + [G.to(DEV)]
        states_a.append((D, a))
        states_b.append(ops.RowState(b[0], b[1] if kind == "adam" else None,
                                     b[2] if kind == "adam" else None, b[3], D, G=b[4]))
    for D, (t, m, v, last, G) in states_a:
        if kind == "adam":
            ops.sparse_adam(t, m, v, last, D, dd, G, scal)
        else:
            ops.sparse_sgd(t, D, dd, G, scal, last_step=last)
    ops.sparse_update_multi(kind, states_b, dd, scal)
    torch.cuda.synchronize()
    for (D, (t, m, v, last, G)), sb in zip(states_a, states_b):
        assert torch.equal(t, sb.table), D
        assert torch.equal(last, sb.last_step), D
        if kind == "adam":
            assert torch.equal(m, sb.m) and torch.equal(v, sb.v), D


def test_pack_columns_multi_all_dtypes():
    g = torch.Generator().manual_seed(0)
    B = 1000
    ids_cols = [torch.randint(0, 1 << 20, (B,), generator=g).double(),
                torch.randint(0, 1 << 20, (B,), generator=g),
                torch.randint(0, 1 << 20, (B, 5), generator=g).int()]
    f_cols = [torch.rand(B, generator=g, dtype=torch.float64), torch.rand(B, generator=g)]
    label = torch.randint(0, 2, (B,), generator=g).double()
    ids = torch.full((B, 9), -1, dtype=torch.int32, device=DEV)
    dense = torch.empty(B, 2, dtype=torch.float32, device=DEV)
    y = torch.empty(B, 1, dtype=torch.float32, device=DEV)
    items, c0 = [], 1
    for t in ids_cols:
        items.append((t.to(DEV), ids, c0))
        c0 += 1 if t.dim() == 1 else t.shape[1]
    items += [(f_cols[0].to(DEV), dense, 0), (f_cols[1].to(DEV), dense, 1), (label.to(DEV), y, 0)]
    ops.pack_columns_multi(items)
    ref = torch.cat([c.reshape(B, -1).to(torch.int32) for c in ids_cols], dim=1)
    assert torch.equal(ids[:, 1:8].cpu(), ref)
    assert int(ids[:, 0].max()) == -1 and int(ids[:, 8].max()) == -1
    assert torch.equal(dense.cpu(), torch.stack([f_cols[0].float(), f_cols[1]], dim=1))
    assert torch.equal(y.cpu().view(-1), label.float())


@pytest.mark.parametrize("series", [False, True])
@pytest.mark.parametrize("t0,lr", [(0, 1e-3), (0, 1e-2), (4000, 1e-3)])
def test_quad_catchup_of_the_deepfm_pair_equals_dense_adam_stepped_k_times(t0, lr, series):
    """fx_catchup_quad (round 5): the D = 16 row and the D = 1 row of one id replayed by the row's quad of lanes.
    The reference steps EVERY row EVERY step (dense torch.optim.Adam, torch_utils.py:72-76 at rank_model.py:322);
    here a row is only brought up to date when a batch next reads it.  Schedule: every row gets gradients now and
    then, with idle gaps of 1 ... 400 steps (past FX_REPLAY_MAX), early in a run (bias corrections moving) and
    4000 steps in; before every touch the caught-up rows must equal the dense run, as must the whole table after
    the final flush — to 2e-6 x max(1, |p|), the bound the plain replays are held to (observed: printed)."""
    from oracle import ctr_oracle as O
    rng = np.random.default_rng(t0 + int(lr * 1e4))
    R = 640
    g0 = torch.Generator().manual_seed(3)
    tabs = {16: torch.randn(R, 16, generator=g0) * 1e-2, 1: torch.randn(R, 1, generator=g0) * 1e-2}
    ws = torch.empty(ops.dedup_workspace_bytes(R), dtype=torch.uint8, device=DEV)
    # row r is touched every period[r] steps (1 ... 400), all rows at the first and at the last step
    period = np.concatenate([np.arange(1, 41), rng.integers(1, 401, R - 40)])
    n_steps = 430
    worst = worst_ref = 0.0
    # ids are rows of tables with R + 1 rows (id 0 = padding)
    R1 = R + 1
    tabs = {D: torch.cat([torch.zeros(1, D), tabs[D]]) for D in (16, 1)}
    ref = {D: (tabs[D].clone(), torch.zeros(R1, D), torch.zeros(R1, D)) for D in (16, 1)}
    # yardstick (round 6): the same trajectory in fp64 — the reference's own fp32 stepping rounds p k times
    r64 = {D: (tabs[D].double(), torch.zeros(R1, D, dtype=torch.float64), torch.zeros(R1, D, dtype=torch.float64))
           for D in (16, 1)}
    dev = {D: [_dev(tabs[D]), torch.zeros(R1, D, device=DEV), torch.zeros(R1, D, device=DEV),
               torch.full((R1,), t0, dtype=torch.int32, device=DEV)] for D in (16, 1)}
    scal = ops.new_scalars(DEV, lr=lr, series=series)
    scal.view(torch.int32)[_lib.SC_STEP] = t0
    for t in range(1, n_steps + 1):
        touched = np.nonzero((t % period == 0) | (t == 1) | (t == n_steps))[0] + 1
        ids = np.zeros((R, 1), dtype=np.int64)
        ids[:len(touched), 0] = touched
        ops.opt_begin_step(scal)
        states = [ops.RowState(*dev[D], D) for D in (16, 1)]
        dd = ops.dedup_catchup(_dev(ids, torch.int32), _dev([0], torch.int64), _dev([R1], torch.int32),
                               _dev([0], torch.int32), ws, states, scal)
        nu = int(dd.n_unique.item())
        rows = dd.uniq_row[:nu].cpu().long()
        # (the padding id's row 0 is listed too whenever a batch has padding: it never gets a gradient)
        assert sorted(set(rows.tolist()) - {0}) == sorted(touched.tolist())
        # what a forward reads now == the dense run after t0 + t - 1 steps
        if t in (2, 7, 41, 200, 399, 400, n_steps) or t % 97 == 0:
            for D in (16, 1):
                e = (dev[D][0][rows.to(DEV)].cpu().double() - r64[D][0][rows]).abs().max().item()
                e_ref = (ref[D][0][rows].double() - r64[D][0][rows]).abs().max().item()
                worst = max(worst, e)
                worst_ref = max(worst_ref, e_ref)
                assert e <= max(2e-6 * max(1.0, ref[D][0].abs().max().item()), 1.5 * e_ref), (t, D, e, e_ref)
        Gs = {}
        for D in (16, 1):
            G = torch.zeros(dd.n_max, D)
            G[:nu] = torch.randn(nu, D, generator=torch.Generator().manual_seed(t * 3 + D)) * 0.1
            G[:nu][rows == 0] = 0.0
            g_dense = torch.zeros(R1, D)
            g_dense[rows] = G[:nu]
            O.adam_dense(ref[D][0], g_dense, ref[D][1], ref[D][2], t0 + t, lr)
            O.adam_dense(r64[D][0], g_dense.double(), r64[D][1], r64[D][2], t0 + t, lr)
            Gs[D] = _dev(G)
        ops.sparse_update_multi("adam", [ops.RowState(*dev[D], D, G=Gs[D]) for D in (16, 1)], dd, scal)
    torch.cuda.synchronize()
    for D in (16, 1):
        assert int(dev[D][3][1:].min()) == t0 + n_steps
        e = (dev[D][0].cpu().double() - r64[D][0]).abs().max().item()
        e_ref = (ref[D][0].double() - r64[D][0]).abs().max().item()
        assert e <= max(2e-6 * max(1.0, ref[D][0].abs().max().item()), 1.5 * e_ref), (D, e, e_ref)
        assert (dev[D][1].cpu() - ref[D][1]).abs().max().item() <= 1e-7 * max(1.0, ref[D][1].abs().max().item()) + 1e-9
        assert (dev[D][2].cpu() - ref[D][2]).abs().max().item() <= 1e-7 * max(1.0, ref[D][2].abs().max().item()) + 1e-9
    print("[quad catch-up] t0 %d lr %g series %s: worst |p - fp64 Adam| before a touch %.2e (torch fp32 stepping: %.2e)"
          % (t0, lr, series, worst, worst_ref))
