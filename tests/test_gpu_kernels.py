"""Per-kernel parity on a real MI355X: every C-ABI entry point against the oracle / a torch-CPU
restatement on the same seeded inputs.  Bit-exact for index and gather work; fp32 tolerances are
written next to each check."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from fuxictr_amd import _lib, ops  # noqa: E402
from oracle import ctr_oracle as O  # noqa: E402

DEV = "cuda:0"


def _dev(x, dtype=None):
    t = torch.as_tensor(x)
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV).contiguous()


def test_pack_columns_all_dtypes():
    g = torch.Generator().manual_seed(0)
    B = 1000
    cols = [torch.randint(0, 1 << 20, (B,), generator=g).double(),
            torch.randint(0, 1 << 20, (B,), generator=g),
            torch.randint(0, 1 << 20, (B, 5), generator=g).int(),
            torch.randint(0, 100, (B,), generator=g).float()]
    out = torch.full((B, 10), -1, dtype=torch.int32, device=DEV)
    ops.pack_columns([c.to(DEV) for c in cols], out, out_col0=1)
    ref = torch.cat([c.reshape(B, -1).to(torch.int32) for c in cols], dim=1)
    assert torch.equal(out[:, 1:9].cpu(), ref)
    assert int(out[:, 0].max()) == -1 and int(out[:, 9].max()) == -1
    f = torch.empty(B, 2, dtype=torch.float32, device=DEV)
    d = torch.rand(B, generator=g, dtype=torch.float64)
    ops.pack_columns([d.to(DEV), cols[3].to(DEV)], f)
    assert torch.equal(f.cpu(), torch.stack([d.float(), cols[3]], dim=1))


@pytest.mark.parametrize("D", [16, 8, 10, 1, 40, 128])
def test_gather_fwd_bit_exact(D):
    g = torch.Generator().manual_seed(D)
    vocabs = [50, 3, 1000, 7, 20011]
    bases = np.concatenate([[0], np.cumsum(vocabs)[:-1]])
    R = int(sum(vocabs))
    B, C, Fd = 777, len(vocabs), 3
    table = torch.randn(R, D, generator=g)
    num_w = torch.randn(Fd, D, generator=g)
    ids = torch.stack([torch.randint(0, v, (B,), generator=g) for v in vocabs], dim=1).int()
    dense = torch.rand(B, Fd, generator=g)
    # slots: interleave numeric and categorical
    slots_c, slots_n = [1, 2, 4, 5, 7], [0, 3, 6]
    n_slots = 8
    out = torch.full((B, n_slots * D), 7.0, device=DEV)
    scal = ops.new_scalars(DEV)
    ops.emb_gather_fwd(_dev(table), D, _dev(ids), _dev(bases, torch.int64), _dev(vocabs, torch.int32),
                       _dev([s * D for s in slots_c], torch.int64), _dev(dense), _dev(num_w),
                       _dev([s * D for s in slots_n], torch.int64), out, scal)
    ref = torch.zeros(B, n_slots, D)
    for c in range(C):
        ref[:, slots_c[c]] = table[ids[:, c].long() + int(bases[c])]
    for j in range(Fd):
        ref[:, slots_n[j]] = dense[:, j:j + 1] * num_w[j]
    assert torch.equal(out.cpu().view(B, n_slots, D), ref)
    assert int(scal.view(torch.int32)[_lib.SC_ERR]) == 0


def test_gather_flags_bad_ids():
    D, B = 16, 64
    table = torch.randn(10, D)
    ids = torch.randint(0, 10, (B, 1)).int()
    ids[5, 0] = 10
    ids[9, 0] = -1
    out = torch.empty(B, D, device=DEV)
    scal = ops.new_scalars(DEV)
    ops.emb_gather_fwd(_dev(table), D, _dev(ids), _dev([0], torch.int64), _dev([10], torch.int32),
                       _dev([0], torch.int64), None, None, None, out, scal)
    assert int(scal.view(torch.int32)[_lib.SC_ERR]) & _lib.FX_FLAG_BAD_ID
    o = out.cpu()
    assert float(o[5].abs().sum()) == 0 and float(o[9].abs().sum()) == 0
    assert torch.equal(o[0], table[ids[0, 0]])


def _dedup(ids, vocabs, pads):
    bases = np.concatenate([[0], np.cumsum(vocabs)[:-1]]).astype(np.int64)
    R = int(sum(vocabs))
    B, C = ids.shape
    ws = torch.empty(ops.dedup_workspace_bytes(B * C), dtype=torch.uint8, device=DEV)
    dd = ops.dedup(_dev(ids, torch.int32), _dev(bases, torch.int64), _dev(vocabs, torch.int32),
                   _dev(pads, torch.int32), R, ws)
    return dd, bases, R


@pytest.mark.parametrize("B", [1, 63, 4096])
def test_dedup_matches_numpy_unique(B):
    rng = np.random.default_rng(B)
    vocabs = [4, 100, 50000, 9]
    pads = [0, 0, -1, 3]
    ids = np.stack([rng.integers(0, v, B) for v in vocabs], axis=1)
    dd, bases, R = _dedup(ids, vocabs, pads)
    keys = ids + bases[None, :]
    valid = np.ones_like(ids, dtype=bool)
    for c, p in enumerate(pads):
        if p >= 0:
            valid[:, c] = ids[:, c] != p
    flat_keys = keys.reshape(-1)
    flat_valid = valid.reshape(-1)
    uniq, counts = np.unique(flat_keys[flat_valid], return_counts=True)
    nu = int(dd.n_unique.item())
    assert nu == len(uniq)
    assert np.array_equal(dd.uniq_row[:nu].cpu().numpy().astype(np.int64) & 0xFFFFFFFF, uniq)
    seg = dd.seg_start[:nu + 1].cpu().numpy().astype(np.int64)
    assert np.array_equal(np.diff(seg), counts)
    pos = dd.sorted_pos.cpu().numpy().astype(np.int64)
    for u in [0, nu // 2, nu - 1]:
        run = pos[seg[u]:seg[u + 1]]
        assert np.all(np.diff(run) > 0)                       # stable: ascending lookup position
        assert np.all(flat_keys[run] == uniq[u])


@pytest.mark.parametrize("B", [1, 777, 4096])
def test_dedup_column_segmented_fast_path(B):
    """columns_sorted=1: same unique rows / runs as the generic path, except that padding rows may
    appear as unique rows whose lookups carry pos = 0xFFFFFFFF (no contribution)."""
    rng = np.random.default_rng(B + 1)
    vocabs = [4, 100, 50000, 9, 3000000]
    pads = [0, 0, -1, 3, 0]
    ids = np.stack([rng.integers(0, v, B) for v in vocabs], axis=1)
    bases = np.concatenate([[0], np.cumsum(vocabs)[:-1]]).astype(np.int64)
    R = int(sum(vocabs))
    C = len(vocabs)
    ws = torch.empty(ops.dedup_workspace_bytes(B * C), dtype=torch.uint8, device=DEV)
    dd = ops.dedup(_dev(ids, torch.int32), _dev(bases, torch.int64), _dev(vocabs, torch.int32),
                   _dev(pads, torch.int32), R, ws, columns_sorted=True)
    nu = int(dd.n_unique.item())
    keys = (ids + bases[None, :]).reshape(-1)
    uniq_all = np.unique(keys)
    got = dd.uniq_row[:nu].cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    assert np.array_equal(got, uniq_all)
    seg = dd.seg_start[:nu + 1].cpu().numpy().astype(np.int64)
    pos = dd.sorted_pos.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    valid = np.ones_like(ids, dtype=bool)
    for c, p in enumerate(pads):
        if p >= 0:
            valid[:, c] = ids[:, c] != p
    flat_valid = valid.reshape(-1)
    for u in range(0, nu, max(1, nu // 50)):
        run = pos[seg[u]:seg[u + 1]]
        live = run[run != 0xFFFFFFFF]
        assert np.all(np.diff(live) > 0)
        assert np.all(keys[live] == uniq_all[u]) and np.all(flat_valid[live])
        expect = int(((keys == uniq_all[u]) & flat_valid).sum())
        assert len(live) == expect
    assert seg[nu] == B * C


def test_dedup_all_padding_and_all_same():
    ids = np.zeros((128, 2), dtype=np.int64)
    dd, _, _ = _dedup(ids, [5, 5], [0, 0])
    assert int(dd.n_unique.item()) == 0 and int(dd.seg_start[0].item()) == 0
    ids = np.full((4096, 1), 2, dtype=np.int64)
    dd, _, _ = _dedup(ids, [5], [0])
    assert int(dd.n_unique.item()) == 1
    assert dd.seg_start[:2].cpu().tolist() == [0, 4096] and int(dd.uniq_row[0]) == 2


@pytest.mark.parametrize("D,vocabs", [(16, [3, 10, 70000, 500]), (1, [3, 10, 70000, 500]),
                                      (10, [40, 7]), (40, [3, 100])])
def test_grad_reduce_matches_index_add(D, vocabs):
    rng = np.random.default_rng(D)
    B, C = 4096, len(vocabs)
    pads = [0] * C
    ids = np.stack([np.minimum((v * rng.random(B) ** 3).astype(np.int64), v - 1) for v in vocabs], 1)
    dd, bases, R = _dedup(ids, vocabs, pads)
    n_slots = C + 1
    dout = torch.randn(B, n_slots * D, generator=torch.Generator().manual_seed(1))
    offs = [(c + 1) * D for c in range(C)]
    G = torch.zeros(dd.n_max, D, device=DEV)
    sq = torch.empty(ops.emb_grad_reduce_partials(dd.n_max, D), device=DEV)
    scr = torch.zeros(ops.emb_grad_reduce_scratch_ints(dd.n_max), dtype=torch.int32, device=DEV)
    ops.emb_grad_reduce(_dev(dout), n_slots * D, _dev(offs, torch.int64), C, D, dd, G, sq, scr)
    assert int(scr[0]) == 0       # the counter is left reset for the next call
    ref = torch.zeros(R, D, dtype=torch.float64)
    d3 = dout.view(B, n_slots, D).double()
    for c in range(C):
        keep = torch.from_numpy(ids[:, c] != 0)
        ref.index_add_(0, torch.from_numpy(ids[:, c] + bases[c])[keep], d3[keep, c + 1])
    nu = int(dd.n_unique.item())
    rows = dd.uniq_row[:nu].cpu().long()
    got = G[:nu].cpu().double()
    err = (got - ref[rows]).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= 2e-6 * max(scale, 1.0), (err, scale)      # fp32 sums of <= 4096 terms
    total = float(sq.double().sum())
    np.testing.assert_allclose(total, float((ref ** 2).sum()), rtol=1e-5)
    # run-to-run determinism
    G2 = torch.zeros_like(G)
    sq2 = torch.empty_like(sq)
    ops.emb_grad_reduce(_dev(dout), n_slots * D, _dev(offs, torch.int64), C, D, dd, G2, sq2, scr)
    assert torch.equal(G[:nu], G2[:nu]) and torch.equal(sq, sq2)


def test_numeric_grad():
    g = torch.Generator().manual_seed(3)
    B, Fd, D, n_slots = 4096, 13, 16, 39
    dout = torch.randn(B, n_slots * D, generator=g)
    dense = torch.rand(B, Fd, generator=g)
    offs = [2 * j * D for j in range(Fd)]
    out = torch.empty(Fd, D, device=DEV)
    ops.emb_numeric_grad(_dev(dout), n_slots * D, _dev(offs, torch.int64), _dev(dense), D, out)
    ref = torch.stack([(dense[:, j:j + 1].double() * dout.view(B, n_slots, D)[:, 2 * j].double()).sum(0)
                       for j in range(Fd)])
    assert (out.cpu().double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()


def _adam_ref_step(p, g, m, v, t, lr):
    O.adam_dense(p, g, m, v, t, lr)


def _within_yardstick(native, ref32, ref64, floor, what):
    """|native - fp64 Adam| <= max(floor, 1.5 x |torch-fp32 Adam - fp64 Adam|): the reference itself rounds p
    at every step (k roundings of up to half an ulp each); the native path is held to being as close to the
    exact trajectory as the reference's own fp32 stepping is, on the same rows, not to a fixed number that one
    rounding sequence happens to meet."""
    e_nat = (native.double() - ref64).abs().max().item()
    e_ref = (ref32.double() - ref64).abs().max().item()
    assert e_nat <= max(floor, 1.5 * e_ref), (what, e_nat, e_ref)
    return e_nat, e_ref


@pytest.mark.parametrize("series", [False, True])
@pytest.mark.parametrize("D", [16, 1, 10])
def test_sparse_adam_exact_mode_equals_dense_adam(D, series):
    """The heart of the 'exact' claim: touched-rows-only updates + catch-up reproduce the reference's dense
    Adam over the WHOLE table (rows idle for > FX_REPLAY_MAX steps included) — with the step-by-step replay
    and (series) with the Adam series table behind the scalar block.  Yardstick: the same trajectory in fp64."""
    rng = np.random.default_rng(5)
    R, lr = 400, 1e-2
    table0 = torch.randn(R, D, generator=torch.Generator().manual_seed(2))
    p_ref, m_ref, v_ref = table0.clone(), torch.zeros(R, D), torch.zeros(R, D)
    p64, m64, v64 = table0.double(), torch.zeros(R, D, dtype=torch.float64), torch.zeros(R, D, dtype=torch.float64)
    table, m, v = _dev(table0), torch.zeros(R, D, device=DEV), torch.zeros(R, D, device=DEV)
    last = torch.zeros(R, dtype=torch.int32, device=DEV)
    scal = ops.new_scalars(DEV, lr=lr, series=series)
    assert (int(scal.view(torch.int32)[_lib.SC_SERIES_TCAP].item()) > 0) == series
    ws = torch.empty(ops.dedup_workspace_bytes(64), dtype=torch.uint8, device=DEV)
    n_steps = 330
    seen = []
    for t in range(1, n_steps + 1):
        # rows 0..9 are touched only at t = 1 and t = 320 (idle > 256 steps); others randomly
        if t in (1, 320):
            ids = np.arange(64) % 10
        else:
            ids = rng.integers(10, R, 64)
        ids = ids.reshape(64, 1)
        ops.opt_begin_step(scal)
        dd = ops.dedup(_dev(ids, torch.int32), _dev([0], torch.int64), _dev([R], torch.int32),
                       _dev([-1], torch.int32), R, ws)
        ops.adam_catchup(table, m, v, last, D, dd, R, -1, scal)
        nu = int(dd.n_unique.item())
        rows = dd.uniq_row[:nu].cpu().long()
        # the rows a forward would read now equal the dense-Adam table after t-1 steps
        if t in (2, 100, 320):
            seen.append(_within_yardstick(table[rows.to(DEV)].cpu(), p_ref[rows], p64[rows], 2e-6, ("read", t)))
        G = torch.zeros(dd.n_max, D)
        G[:nu] = torch.randn(nu, D, generator=torch.Generator().manual_seed(t)) * 0.1
        g_dense = torch.zeros(R, D)
        g_dense[rows] = G[:nu]
        _adam_ref_step(p_ref, g_dense, m_ref, v_ref, t, lr)
        O.adam_dense(p64, g_dense.double(), m64, v64, t, lr)
        ops.sparse_adam(table, m, v, last, D, dd, _dev(G), scal)
    ops.adam_catchup(table, m, v, last, D, None, R, 0, scal)       # flush
    assert int(last.min()) == n_steps
    seen.append(_within_yardstick(table.cpu(), p_ref, p64, 2e-6, "flush"))
    assert (m.cpu() - m_ref).abs().max().item() <= 1e-6
    assert (v.cpu() - v_ref).abs().max().item() <= 1e-6
    print("[exact mode] D %d series %s: |native - fp64| / |torch fp32 - fp64| at t = 2, 100, 320, flush: %s"
          % (D, series, ", ".join("%.1e / %.1e" % s for s in seen)))


def test_sparse_sgd_and_clip():
    R, D = 50, 16
    table0 = torch.randn(R, D)
    table = _dev(table0)
    scal = ops.new_scalars(DEV, lr=0.5, max_norm=1.0)
    ids = np.array([[3], [7], [3], [9]])
    ws = torch.empty(ops.dedup_workspace_bytes(4), dtype=torch.uint8, device=DEV)
    dd = ops.dedup(_dev(ids, torch.int32), _dev([0], torch.int64), _dev([R], torch.int32),
                   _dev([-1], torch.int32), R, ws)
    G = torch.zeros(4, D)
    G[:3] = torch.randn(3, D)
    sq = _dev([(G ** 2).sum().item()], torch.float32)
    ops.clip_coef([sq], scal)
    total = G.norm().item()
    coef = min(1.0, 1.0 / (total + 1e-6))
    np.testing.assert_allclose(float(scal[_lib.SC_CLIP]), coef, rtol=1e-6)
    np.testing.assert_allclose(float(scal[_lib.SC_TOTAL_NORM]), total, rtol=1e-6)
    ops.sparse_sgd(table, D, dd, _dev(G), scal)
    ref = table0.clone()
    ref[[3, 7, 9]] -= 0.5 * coef * G[:3]
    assert (table.cpu() - ref).abs().max().item() <= 1e-6


def test_multi_tensor_adam_sqnorm_clip():
    g = torch.Generator().manual_seed(9)
    shapes = [(1024, 624), (1024,), (1, 1024), (1,), (13, 16), (37,)]
    ps = [torch.randn(*s, generator=g) for s in shapes]
    gs = [torch.randn(*s, generator=g) for s in shapes]
    dp = [_dev(p) for p in ps]
    dg = [_dev(x) for x in gs]
    dm = [torch.zeros_like(p) for p in dp]
    dv = [torch.zeros_like(p) for p in dp]
    rm = [torch.zeros_like(p) for p in ps]
    rv = [torch.zeros_like(p) for p in ps]
    scal = ops.new_scalars(DEV, lr=1e-3, max_norm=10.0)
    sq = torch.empty(len(ps) * _lib.FX_MT_BLOCKS, device=DEV)
    for t in range(1, 4):
        ops.opt_begin_step(scal)
        ops.mt_sqnorm(dg, sq)
        ops.clip_coef([sq], scal)
        ops.mt_adam(dp, dg, dm, dv, scal)
        grads = [x.clone() for x in gs]
        truth = float(torch.sqrt(sum((x.double() ** 2).sum() for x in gs)))
        total = O.clip_grad_norm(grads, 10.0)
        # native: fp32 partials, fp64 final sum -> within 1e-6 of the fp64 truth; torch's CPU
        # vector_norm (the reference's arithmetic) is itself only ~1e-5 accurate at this size
        np.testing.assert_allclose(float(scal[_lib.SC_TOTAL_NORM]), truth, rtol=1e-6)
        np.testing.assert_allclose(total, truth, rtol=2e-5)
        for p, x, m, v in zip(ps, grads, rm, rv):
            O.adam_dense(p, x, m, v, t, 1e-3)
    for a, b in zip(dp, ps):
        assert (a.cpu() - b).abs().max().item() <= 2e-6
    ops.mt_sgd(dp, dg, scal)
    coef = float(scal[_lib.SC_CLIP])
    for a, b, x in zip(dp, ps, gs):
        assert (a.cpu() - (b - 1e-3 * coef * x)).abs().max().item() <= 2e-6


@pytest.mark.parametrize("F,D", [(39, 16), (19, 8), (12, 10), (5, 40)])
def test_fm_forward_backward(F, D):
    g = torch.Generator().manual_seed(F)
    B = 517
    emb = torch.randn(B, F, D, generator=g) * 0.3
    add = torch.randn(B, 1, generator=g)
    out = torch.empty(B, 1, device=DEV)
    ops.fm_fwd(_dev(emb).view(B, F * D), F, D, _dev(add), out)
    e = emb.clone().requires_grad_(True)
    ref = O.fm_product_sum(e) + add
    assert (out.cpu() - ref.detach()).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())
    gout = torch.randn(B, 1, generator=g)
    ref.backward(gout)
    demb = torch.empty(B, F * D, device=DEV)
    ops.fm_bwd(_dev(emb).view(B, F * D), F, D, _dev(gout), demb)
    assert (demb.cpu().view(B, F, D) - e.grad).abs().max().item() <= 1e-5 * e.grad.abs().max().item()
    demb2 = torch.ones(B, F * D, device=DEV)
    ops.fm_bwd(_dev(emb).view(B, F * D), F, D, _dev(gout), demb2, accumulate=True)
    assert (demb2 - 1 - demb).abs().max().item() <= 1e-5


def test_lr_forward():
    g = torch.Generator().manual_seed(4)
    vocabs = [50, 3, 1000, 7] * 6 + [11, 13]
    bases = np.concatenate([[0], np.cumsum(vocabs)[:-1]])
    B, C, Fd = 1000, len(vocabs), 13
    table = torch.randn(int(sum(vocabs)), 1, generator=g)
    ids = torch.stack([torch.randint(0, v, (B,), generator=g) for v in vocabs], dim=1).int()
    dense = torch.rand(B, Fd, generator=g)
    w1 = torch.randn(Fd, 1, generator=g)
    bias = torch.randn(1, generator=g)
    out = torch.empty(B, 1, device=DEV)
    scal = ops.new_scalars(DEV)
    ops.lr_fwd(_dev(table), _dev(ids), _dev(bases, torch.int64), _dev(vocabs, torch.int32),
               _dev(dense), _dev(w1), _dev(bias), out, scal)
    ref = sum(table[ids[:, c].long() + int(bases[c])] for c in range(C)) + dense @ w1 + bias
    assert (out.cpu() - ref).abs().max().item() <= 1e-5


GEMM_SHAPES = [(4096, 1024, 624), (4096, 1024, 1024), (4096, 1, 1024), (100, 70, 50), (257, 129, 17),
               (1, 1, 1), (64, 624, 4096), (130, 1648, 33),
               # rows only 4-byte aligned (DLRM: 367 = 27*26/2 + 16 inputs of the top MLP): the
               # pipelined kernel's UA loaders — straddling float4 at the end of K / of the rows
               (4096, 1024, 367), (4096, 367, 1024), (367, 1024, 4096), (333, 65, 37), (66, 67, 9),
               (5, 7, 11), (4099, 130, 131),
               # the Linear(hidden -> 1) head of a tower: forward N = 1 (row dots), input gradient K = 1
               # (outer product, 16-byte stores), also with small odd extents
               (4096, 1024, 1), (300, 64, 3), (77, 1648, 1), (4096, 2, 2048), (513, 3, 40)]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("ta,tb", [(False, True), (False, False), (True, False), (True, True)])
def test_gemm_all_layouts(M, N, K, ta, tb):
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn((K, M) if ta else (M, K), generator=g)
    Bm = torch.randn((N, K) if tb else (K, N), generator=g)
    C = torch.full((M, N), float("nan"), device=DEV)
    ops.gemm(_dev(A), _dev(Bm), C, transa=ta, transb=tb)
    a = A.t() if ta else A
    b = Bm.t() if tb else Bm
    ref = a.double() @ b.double()
    bound = (a.abs().double() @ b.abs().double()).max().item()
    err = (C.cpu().double() - ref).abs().max().item()
    assert err <= 2e-6 * bound, (err, bound)             # fp32 accumulate over K terms


@pytest.mark.parametrize("M,N,K", [(4096, 1024, 1024), (4096, 1024, 624), (4096, 624, 624),
                                   (2048, 512, 128), (300, 70, 52), (257, 33, 9)])
def test_gemm_dw_dx_pair_is_bit_identical_to_the_two_gemms(M, N, K):
    """fx_gemm_f32_batch: the weight- and input-gradient products of a layer as ONE grid (the aligned
    tower shapes: 128-row tiles, two workgroups per CU, tiles and K slabs chosen per launch) or problem
    by problem (ragged shapes), with the fused bias gradient, the ReLU mask and the residual add.
    dX (no K split: one k-ordered chain per element) carries the same bits as fx_gemm_f32; dW may be
    split differently, so it is compared with fp64 within the fp32 bound; two runs are bit-identical."""
    g = torch.Generator().manual_seed(M + N + K)
    dz, x = _dev(torch.randn(M, N, generator=g)), _dev(torch.randn(M, K, generator=g))
    W = _dev(torch.randn(N, K, generator=g) * 0.1)
    mask, add = _dev(torch.randn(M, K, generator=g)), _dev(torch.randn(M, K, generator=g))
    sk = 8 if M >= 2048 else 2
    ws = torch.empty(ops.gemm_workspace_floats(N, K, sk), device=DEV)
    out = []
    for pair in (True, True, False):
        dW = torch.full((N, K), float("nan"), device=DEV)
        dx = torch.full((M, K), float("nan"), device=DEV)
        db = torch.full((N,), float("nan"), device=DEV)
        if pair:
            ops.gemm_dw_dx(dz, x, W, dW, dx, split_k=sk, workspace=ws, rowsum=db, mask=mask, add=add)
        else:
            ops.gemm(dz, x, dW, transa=True, transb=False, split_k=min(sk, 4), workspace=ws, rowsum=db)
            ops.gemm(dz, W, dx, transa=False, transb=False, mask=mask, add=add)
        out.append((dW, dx, db))
    for a, b in zip(out[0], out[1]):
        assert torch.equal(a, b)                         # deterministic
    assert torch.equal(out[0][1], out[2][1])             # dX: same bits as the single GEMM
    for dW, dx, db in (out[0], out[2]):
        ref_w = dz.double().t() @ x.double()
        assert (dW.double() - ref_w).abs().max().item() <= 3e-6 * (dz.abs().double().t() @ x.abs().double()).max().item()
        ref_x = torch.where(mask > 0, dz.double() @ W.double(), torch.zeros((), dtype=torch.float64, device=DEV)) + add.double()
        assert (dx.double() - ref_x).abs().max().item() <= 3e-6 * (dz.abs().double() @ W.abs().double()).max().item() + 1e-6
        assert (db.double() - dz.double().sum(0)).abs().max().item() <= 1e-5 * dz.abs().double().sum(0).max().item()


def test_gemm_batch_of_four_problems_in_one_grid():
    """fx_gemm_f32_batch with four problems (DCNv2's parallel structure: the cross and the deep layer of
    one depth, dW + dX each) == the same four products as single launches, within the fp32 bound."""
    g = torch.Generator().manual_seed(11)
    M = 4096
    items = []
    for N, K in ((624, 624), (1024, 624)):
        dz, x = _dev(torch.randn(M, N, generator=g)), _dev(torch.randn(M, K, generator=g))
        W = _dev(torch.randn(N, K, generator=g) * 0.1)
        items.append((dz, x, W))
    probs, outs, keep = [], [], []
    for dz, x, W in items:
        N, K = W.shape
        dW, dx = torch.full((N, K), float("nan"), device=DEV), torch.full((M, K), float("nan"), device=DEV)
        db = torch.full((N,), float("nan"), device=DEV)
        ws = torch.empty(ops.gemm_workspace_floats(N, K, 8), device=DEV)
        probs.append(ops.gemm_problem(dz, x, dW, transa=True, transb=False, split_k=8, workspace=ws,
                                      rowsum=db))
        probs.append(ops.gemm_problem(dz, W, dx, transa=False, transb=False))
        outs.append((dW, dx, db))
        keep.append(ws)
    ops.gemm_batch(probs)
    torch.cuda.synchronize()
    for (dz, x, W), (dW, dx, db) in zip(items, outs):
        ref_w = dz.double().t() @ x.double()
        assert (dW.double() - ref_w).abs().max().item() <= 3e-6 * (dz.abs().double().t() @ x.abs().double()).max().item()
        ref_x = dz.double() @ W.double()
        assert (dx.double() - ref_x).abs().max().item() <= 3e-6 * (dz.abs().double() @ W.abs().double()).max().item()
        assert (db.double() - dz.double().sum(0)).abs().max().item() <= 1e-5 * dz.abs().double().sum(0).max().item()


@pytest.mark.parametrize("B,K,masked", [(4096, 1024, True), (1000, 64, True), (333, 256, False), (4096, 368, True)])
def test_head_backward_in_one_pass_is_the_two_skinny_launches(B, K, masked):
    """gemm_dw_dx for a Linear(hidden -> 1) head (k_head_bwd_v4: dW, db and the masked dX in one pass over
    the activations) == the two skinny launches it replaces, bit for bit (same slab order for dW, exact
    single products for dX)."""
    g = torch.Generator().manual_seed(B + K)
    h = _dev(torch.relu(torch.randn(B, K, generator=g)))          # a ReLU layer's output: its own mask
    dz = _dev(torch.randn(B, 1, generator=g))
    W = _dev(torch.randn(1, K, generator=g))
    sk = 16
    res = []
    for fused in (True, False):
        dW = torch.full((1, K), float("nan"), device=DEV)
        dx = torch.full((B, K), float("nan"), device=DEV)
        db = torch.full((1,), float("nan"), device=DEV)
        ws = torch.empty(ops.gemm_workspace_floats(1, K, sk), device=DEV)
        mask = h if masked else None
        if fused:
            ops.gemm_dw_dx(dz, h, W, dW, dx, split_k=sk, workspace=ws, rowsum=db, mask=mask)
        else:
            ops.gemm(dz, h, dW, transa=True, transb=False, split_k=sk, workspace=ws, rowsum=db)
            ops.gemm(dz, W, dx, transa=False, transb=False, mask=mask)
        torch.cuda.synchronize()
        res.append((dW.cpu(), dx.cpu(), db.cpu()))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    ref_w = dz.double().t() @ h.double()
    assert (res[0][0].double() - ref_w.cpu()).abs().max().item() <= 1e-5 * (dz.abs().double().t() @ h.abs().double()).max().item()
    ref_x = dz.double() @ W.double()
    if masked:
        ref_x = ref_x * (h > 0).double()
    assert (res[0][1].double() - ref_x.cpu()).abs().max().item() <= 1e-6 * ref_x.abs().max().item()


@pytest.mark.parametrize("M", [4096, 1000])
def test_gemm_batch_of_two_forward_problems_is_the_two_launches(M):
    """fx_gemm_f32_batch with the cross layer and the deep layer of one DCNv2 depth FORWARD (x W^T with the
    CrossNet epilogue bias / z out / Hadamard / residual, and bias + ReLU): one grid of 64x64 tiles, same
    tiles and k order as the two single launches -> identical bits, strided outputs included."""
    g = torch.Generator().manual_seed(M)
    D0, H = 624, 1024
    xi, x0 = _dev(torch.randn(M, D0, generator=g)), _dev(torch.randn(M, D0, generator=g))
    Wc, bc = _dev(torch.randn(D0, D0, generator=g) * 0.05), _dev(torch.randn(D0, generator=g))
    Wd, bd = _dev(torch.randn(H, D0, generator=g) * 0.05), _dev(torch.randn(H, generator=g))
    res = []
    for batched in (True, False):
        out = torch.full((M, D0 + H), float("nan"), device=DEV)      # both towers write one buffer
        z = torch.full((M, D0), float("nan"), device=DEV)
        if batched:
            ops.gemm_batch([ops.gemm_problem(xi, Wc, out[:, :D0], transb=True, bias=bc, zout=z, mul=x0, add=xi),
                            ops.gemm_problem(xi, Wd, out[:, D0:], transb=True, bias=bd, act=1)])
        else:
            ops.gemm(xi, Wc, out[:, :D0], transb=True, bias=bc, zout=z, mul=x0, add=xi)
            ops.gemm(xi, Wd, out[:, D0:], transb=True, bias=bd, act=1)
        torch.cuda.synchronize()
        res.append((out.cpu(), z.cpu()))
    if M == 4096:
        # (both forms on the same kernels — round 5: the split-bf16 ones, every problem has >= 96 tiles)
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    # M = 1000: the batch (104 tiles of 128x128) leaves on the split-bf16 kernels, the single launches (40 and
    # 64 tiles) stay on the fp32-MFMA ones — same products, different rounding: compared through fp64 below
    zr = xi.double() @ Wc.double().t() + bc.double()
    ref = torch.cat([zr * x0.double() + xi.double(),
                     torch.relu(xi.double() @ Wd.double().t() + bd.double())], dim=1).cpu()
    for r in res:
        assert (r[0].double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


@pytest.mark.parametrize("widths", [(16, 1), (16,), (8, 4, 1), (3,)])
def test_exchange_block_scatter_and_split(widths):
    """fx_scatter_rows into the column range of a shared block (row stride > D, 16-byte aligned or
    not) and fx_split_rows back into per-group row buffers with a zero pad row: bit-exact moves."""
    g = torch.Generator().manual_seed(sum(widths))
    n_slots, n_max = 5000, 3000
    W = sum(widths) if len(widths) == 1 else -(-sum(widths) // 4) * 4
    block = torch.zeros(n_slots + 1, W, device=DEV)
    ref = torch.zeros(n_slots + 1, W)
    n_rows = torch.tensor([2777], dtype=torch.int32)
    off = 0
    for D in widths:
        src = torch.randn(n_max, D, generator=g)
        row_map = torch.randperm(n_slots, generator=g)[:n_max].int()
        ops.scatter_rows(_dev(src), _dev(row_map), _dev(n_rows), n_max, D, block[:, off:off + D])
        ref[row_map[:2777].long(), off:off + D] = src[:2777]
        off += D
    assert torch.equal(block.cpu(), ref)
    parts, off = [], 0
    for D in widths:
        parts.append((off, torch.full((n_slots + 1, D), float("nan"), device=DEV)))
        off += D
    ops.split_rows(block, n_slots, parts, zero_tail_rows=1)
    for o, dst in parts:
        D = dst.shape[1]
        assert torch.equal(dst[:n_slots].cpu(), ref[:n_slots, o:o + D])
        assert float(dst[n_slots].abs().max()) == 0.0


def test_gemm_unaligned_rows_with_epilogues_split_k_and_views():
    """The UA path with everything on: operands that are column slices of wider buffers (4-byte
    aligned base pointers and odd leading dimensions), split-K with the fused row sums, bias + ReLU +
    mask + add epilogues."""
    g = torch.Generator().manual_seed(77)
    M, N, K = 1030, 367, 1501
    big = torch.randn(M, K + 5, generator=g)
    Wb = torch.randn(N, K + 3, generator=g)
    A, W = _dev(big)[:, 3:3 + K], _dev(Wb)[:, 1:1 + K]
    bias, add, msk = _dev(torch.randn(N, generator=g)), _dev(torch.randn(M, N, generator=g)), _dev(torch.randn(M, N, generator=g))
    ref = A.double() @ W.double().t() + bias.double()
    tol = 3e-6 * (A.abs().double() @ W.abs().double().t()).max().item()
    C = torch.empty(M, N, device=DEV)
    ops.gemm(A, W, C, transb=True, bias=bias, act=1, mask=msk, add=add)
    want = torch.where(msk > 0, ref.clamp(min=0), torch.zeros((), dtype=torch.float64, device=DEV)) + add.double()
    assert (C.double() - want).abs().max().item() <= tol
    # weight-gradient layout: dW[N, K] = dz^T x with dz [M, N] and x a strided [M, K] view, split-K 3
    dz = _dev(torch.randn(M, N, generator=g))
    dW = torch.empty(N, K, device=DEV)
    db = torch.empty(N, device=DEV)
    ws = torch.empty(ops.gemm_workspace_floats(N, K, 3), device=DEV)
    ops.gemm(dz, A, dW, transa=True, transb=False, split_k=3, workspace=ws, rowsum=db)
    refw = dz.double().t() @ A.double()
    assert (dW.double() - refw).abs().max().item() <= 3e-6 * (dz.abs().double().t() @ A.abs().double()).max().item()
    assert (db.double() - dz.double().sum(0)).abs().max().item() <= 1e-5 * dz.abs().double().sum(0).max().item()
    # input-gradient layout: dx[M, K] = dz W into a strided view
    out = torch.zeros(M, K + 2, device=DEV)
    ops.gemm(dz, W, out[:, 1:1 + K], transa=False, transb=False)
    refx = dz.double() @ W.double()
    assert (out[:, 1:1 + K].double() - refx).abs().max().item() <= 3e-6 * (dz.abs().double() @ W.abs().double()).max().item()
    assert float(out[:, 0].abs().max()) == 0.0 and float(out[:, K + 1].abs().max()) == 0.0


def test_gemm_head_layer_input_gradient_with_relu_mask():
    """dX of the head: dz [M,1] x W [1,N] with the ReLU mask of the layer below in the epilogue."""
    g = torch.Generator().manual_seed(3)
    M, N = 4096, 1024
    dz, W = _dev(torch.randn(M, 1, generator=g)), _dev(torch.randn(1, N, generator=g))
    h = _dev(torch.randn(M, N, generator=g))
    dx = torch.empty(M, N, device=DEV)
    ops.gemm(dz, W, dx, transa=False, transb=False, mask=h)
    ref = torch.where(h > 0, dz * W, torch.zeros((), device=DEV))
    assert torch.equal(dx, ref)


def test_gemm_mfma_layout_is_not_transposed():
    """A = I with an ASYMMETRIC B catches a swapped C/D fragment mapping."""
    n = 128
    A = torch.eye(n)
    Bm = torch.arange(n * n, dtype=torch.float32).view(n, n) / 100.0
    C = torch.empty(n, n, device=DEV)
    ops.gemm(_dev(A), _dev(Bm), C)
    assert torch.equal(C.cpu(), Bm)


def test_gemm_epilogues_and_split_k():
    g = torch.Generator().manual_seed(21)
    M, N, K = 515, 200, 3000
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g)
    bias = torch.randn(N, generator=g)
    mul = torch.randn(M, N, generator=g)
    add = torch.randn(M, N, generator=g)
    msk = torch.randn(M, N, generator=g)
    z_ref = (A.double() @ W.double().t() + bias.double())
    tol = 3e-6 * (A.abs().double() @ W.abs().double().t()).max().item()
    for sk in (1, 5):
        ws = torch.empty(sk * M * N, device=DEV)
        C = torch.empty(M, N, device=DEV)
        Z = torch.empty(M, N, device=DEV)
        ops.gemm(_dev(A), _dev(W), C, transb=True, bias=_dev(bias), act=1, split_k=sk, workspace=ws)
        assert (C.cpu().double() - z_ref.clamp(min=0)).abs().max().item() <= tol
        ops.gemm(_dev(A), _dev(W), C, transb=True, bias=_dev(bias), zout=Z, mul=_dev(mul),
                 add=_dev(add), split_k=sk, workspace=ws)
        assert (Z.cpu().double() - z_ref).abs().max().item() <= tol
        assert (C.cpu().double() - (z_ref * mul.double() + add.double())).abs().max().item() <= 3 * tol
        ops.gemm(_dev(A), _dev(W), C, transb=True, mask=_dev(msk), split_k=sk, workspace=ws)
        ref = torch.where(msk > 0, (A.double() @ W.double().t()), torch.zeros((), dtype=torch.float64))
        assert (C.cpu().double() - ref).abs().max().item() <= tol


@pytest.mark.parametrize("M,N,K,sk", [(1024, 624, 4096, 6), (1024, 1024, 4096, 1), (1, 1024, 4096, 64),
                                      (100, 70, 515, 3), (32, 1648, 4096, 16)])
def test_gemm_fused_rowsum_is_the_bias_gradient(M, N, K, sk):
    g = torch.Generator().manual_seed(M + K)
    dz = torch.randn(K, M, generator=g)          # [batch, N_out]
    x = torch.randn(K, N, generator=g)           # [batch, K_in]
    dW = torch.empty(M, N, device=DEV)
    db = torch.full((M,), float("nan"), device=DEV)
    ws = torch.empty(ops.gemm_workspace_floats(M, N, sk), device=DEV)
    ops.gemm(_dev(dz), _dev(x), dW, transa=True, split_k=sk, workspace=ws, rowsum=db)
    ref = dz.double().t() @ x.double()
    assert (dW.cpu().double() - ref).abs().max().item() <= 3e-6 * (dz.abs().double().t() @ x.abs().double()).max().item()
    refb = dz.double().sum(0)
    assert (db.cpu().double() - refb).abs().max().item() <= 1e-6 * dz.abs().double().sum(0).max().item()


def test_colsum_mask_cross_prep_bce():
    g = torch.Generator().manual_seed(8)
    M, N = 4096, 1000
    X = torch.randn(M, N, generator=g)
    out = torch.empty(N, device=DEV)
    ws = torch.empty(_lib.FX_COLSUM_CHUNKS * N, device=DEV)
    ops.colsum(_dev(X), out, ws)
    assert (out.cpu().double() - X.double().sum(0)).abs().max().item() <= 2e-4
    Y = torch.randn(M, N, generator=g)
    o = ops.mask_mul(_dev(X), _dev(Y), torch.empty(M, N, device=DEV))
    assert torch.equal(o.cpu(), torch.where(Y > 0, X, torch.zeros(())))
    Z = torch.randn(M, N, generator=g)
    t = torch.empty(M, N, device=DEV)
    dx0 = torch.ones(M, N, device=DEV)
    ops.cross_bwd_prep(_dev(X), _dev(Y), _dev(Z), t, dx0, init=False, add_dxn=True)
    assert torch.allclose(t.cpu(), X * Y, atol=0, rtol=0)
    assert (dx0.cpu() - (1 + X * Z + X)).abs().max().item() <= 1e-5
    ops.cross_bwd_prep(_dev(X), _dev(Y), _dev(Z), t, dx0, init=True, add_dxn=False)
    assert torch.equal(dx0.cpu(), X * Z)
    # the incoming gradient as a column slice of a wider tensor (backward of a torch.cat): read in place
    wide = torch.randn(M, N + 24, generator=g)
    dw = _dev(wide)[:, 17:17 + N]
    o = ops.mask_mul(dw, _dev(Y), torch.empty(M, N, device=DEV))
    assert torch.equal(o.cpu(), torch.where(Y > 0, wide[:, 17:17 + N], torch.zeros(())))
    ops.cross_bwd_prep(dw, _dev(Y), _dev(Z), t, dx0, init=True, add_dxn=True)
    assert torch.equal(t.cpu(), wide[:, 17:17 + N] * Y)
    assert (dx0.cpu() - (wide[:, 17:17 + N] * Z + wide[:, 17:17 + N])).abs().max().item() <= 1e-6
    # sigmoid + BCE: value and gradient against torch autograd on the reference's two ops
    B = 4096
    logit = (torch.randn(B, 1, generator=g) * 4).requires_grad_(True)
    logit.data[0] = 40.0
    logit.data[1] = -40.0       # exercise the log clamp at -100
    y = (torch.rand(B, 1, generator=g) < 0.3).float()
    y[0] = 0.0
    y[1] = 1.0
    p_ref = torch.sigmoid(logit)
    loss_ref = O.bce_mean(p_ref, y)
    loss_ref.backward()
    prob = torch.empty(B, 1, device=DEV)
    loss = torch.empty((), device=DEV)
    dl = torch.empty(B, 1, device=DEV)
    ops.sigmoid_bce(_dev(logit.detach()), _dev(y), prob=prob, loss=loss, dlogit=dl)
    assert (prob.cpu() - p_ref.detach()).abs().max().item() <= 2e-7
    np.testing.assert_allclose(float(loss), float(loss_ref.detach()), rtol=2e-6)
    assert (dl.cpu() - logit.grad).abs().max().item() <= 1e-9 + 2e-6 * logit.grad.abs().max().item()
    ops.sigmoid_bce(_dev(logit.detach()), None, prob=prob)
    assert (prob.cpu() - p_ref.detach()).abs().max().item() <= 2e-7


# ---- DIN attention pieces and Dice -----------------------------------------------------------
def _seq_inputs(B=333, L=7, E=16, seed=0):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(B, E, generator=g)
    rec = torch.randn(B, L + 3, E, generator=g)        # K is a strided view of a wider record
    K = rec[:, 2:2 + L, :]
    ids = torch.randint(0, 4, (B, L), generator=g).int()
    return q, rec, K, ids


def test_din_concat_and_pool_match_autograd():
    q, rec, K, ids = _seq_inputs()
    B, L, E = K.shape
    dq_, drec = _dev(q), _dev(rec)
    dK_ = drec[:, 2:2 + L, :]
    x = torch.empty(B * L, 4 * E, device=DEV)
    ops.din_concat_fwd(dq_, dK_, x)
    qr, Kr = q.clone().requires_grad_(True), K.clone().requires_grad_(True)
    t = qr.unsqueeze(1).expand(-1, L, -1)
    ref = torch.cat([t, Kr, t - Kr, t * Kr], dim=-1).view(B * L, 4 * E)
    assert torch.equal(x.cpu(), ref.detach())
    gx = torch.randn(B * L, 4 * E, generator=torch.Generator().manual_seed(1))
    ref.backward(gx)
    dq = torch.empty(B, E, device=DEV)
    dK = torch.empty(B, L, E, device=DEV)
    ops.din_concat_bwd(_dev(gx), dq_, dK_, dq, dK)
    assert (dq.cpu() - qr.grad).abs().max().item() <= 1e-5
    assert (dK.cpu() - Kr.grad).abs().max().item() <= 1e-5
    # pooling
    w = torch.randn(B, L, generator=torch.Generator().manual_seed(2))
    out = torch.empty(B, E, device=DEV)
    ops.din_pool_fwd(_dev(w), _dev(ids), dK_, out)
    wr, Kr2 = w.clone().requires_grad_(True), K.clone().requires_grad_(True)
    refo = ((wr * (ids != 0).float()).unsqueeze(-1) * Kr2).sum(1)
    assert (out.cpu() - refo.detach()).abs().max().item() <= 1e-5
    go = torch.randn(B, E, generator=torch.Generator().manual_seed(3))
    refo.backward(go)
    dw = torch.empty(B, L, device=DEV)
    dK2 = torch.empty(B, L, E, device=DEV)
    ops.din_pool_bwd(_dev(w), _dev(ids), dK_, _dev(go), dw, dK2)
    assert (dw.cpu() - wr.grad).abs().max().item() <= 1e-5
    assert (dK2.cpu() - Kr2.grad).abs().max().item() <= 1e-5


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("N,H", [(20480, 64), (777, 16), (5, 100)])
def test_dice_matches_reference_batchnorm_gate(training, N, H):
    g = torch.Generator().manual_seed(N + H)
    z = (torch.randn(N, H, generator=g) * 1.5 + 0.3)
    alpha = torch.rand(H, generator=g) - 0.5
    rm0 = torch.randn(H, generator=g) * 0.1
    rv0 = torch.rand(H, generator=g) + 0.5
    state = {"p.bn.running_mean": rm0.clone(), "p.bn.running_var": rv0.clone(),
             "p.bn.num_batches_tracked": torch.zeros((), dtype=torch.long),
             "p.alpha": alpha.clone().requires_grad_(True)}
    zr = z.clone().requires_grad_(True)
    ref = O.dice(state, "p.", zr, training)
    gy = torch.randn(N, H, generator=g)
    ref.backward(gy)
    rm, rv = _dev(rm0), _dev(rv0)
    stats = torch.empty(2 * H, device=DEV)
    y = torch.empty(N, H, device=DEV)
    ws = torch.empty(ops.dice_workspace_floats(H), device=DEV)
    ops.dice_fwd(_dev(z), _dev(alpha), 1e-9, 0.01, training, rm, rv, stats, y, ws)
    assert (y.cpu() - ref.detach()).abs().max().item() <= 2e-5
    assert (rm.cpu() - state["p.bn.running_mean"]).abs().max().item() <= 1e-6
    assert (rv.cpu() - state["p.bn.running_var"]).abs().max().item() <= 1e-5
    dz = torch.empty(N, H, device=DEV)
    da = torch.empty(H, device=DEV)
    ops.dice_bwd(_dev(z), _dev(gy), _dev(alpha), 1e-9, training, stats, dz, da, ws)
    scale = max(1.0, zr.grad.abs().max().item())
    assert (dz.cpu() - zr.grad).abs().max().item() <= 5e-5 * scale
    assert (da.cpu() - state["p.alpha"].grad).abs().max().item() <= 2e-4 * max(1.0, state["p.alpha"].grad.abs().max().item())


@pytest.mark.parametrize("F,D", [(27, 16), (15, 8), (2, 3), (40, 40), (32, 32), (31, 2), (3, 4), (2, 16)])
def test_dot_interaction_matches_bmm_triu(F, D):
    g = torch.Generator().manual_seed(F * D)
    B = 300
    emb = torch.randn(B, F, D, generator=g)
    out = torch.empty(B, F * (F - 1) // 2, device=DEV)
    ops.dot_interact_fwd(_dev(emb).view(B, F * D), F, D, out)
    e = emb.clone().requires_grad_(True)
    ref = O.dot_interaction(e)
    assert (out.cpu() - ref.detach()).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy)
    demb = torch.empty(B, F * D, device=DEV)
    ops.dot_interact_bwd(_dev(emb).view(B, F * D), _dev(gy), F, D, demb)
    assert (demb.cpu().view(B, F, D) - e.grad).abs().max().item() <= 2e-5 * max(1.0, e.grad.abs().max().item())


@pytest.mark.parametrize("F,D,pad", [(27, 16, 1), (15, 8, 3), (5, 4, 0), (27, 16, 0), (40, 40, 2)])
def test_dot_interaction_with_the_last_field_appended(F, D, pad):
    """tail mode of fx_dot_interact_fwd / _bwd (DLRM's [dots | dense vector | zero padding] row and its
    gradient, both kernels' MFMA and workgroup-per-sample forms): == the unfused composition."""
    g = torch.Generator().manual_seed(F + D + pad)
    B, P = 333, F * (F - 1) // 2
    emb = torch.randn(B, F, D, generator=g)
    out = torch.full((B, P + D + pad), float("nan"), device=DEV)
    ops.dot_interact_fwd(_dev(emb).view(B, F * D), F, D, out, tail=D + pad)
    e = emb.clone().requires_grad_(True)
    ref = torch.cat([O.dot_interaction(e), e[:, F - 1, :], torch.zeros(B, pad)], dim=1)
    assert (out.cpu() - ref.detach()).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())
    assert torch.equal(out.cpu()[:, P:P + D], emb[:, F - 1, :]) and (out.cpu()[:, P + D:] == 0).all()
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy)
    demb = torch.full((B, F * D), float("nan"), device=DEV)
    ops.dot_interact_bwd(_dev(emb).view(B, F * D), _dev(gy), F, D, demb, tail=D + pad)
    assert (demb.cpu().view(B, F, D) - e.grad).abs().max().item() <= 2e-5 * max(1.0, e.grad.abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize("F0,D,units,B", [(39, 16, [16, 16, 16], 700), (9, 8, [12, 6, 5], 130),
                                          (5, 10, [7], 3), (26, 40, [8, 4], 260),
                                          # D = 16, O <= 16: the MFMA kernels (ragged O, Mi, B)
                                          (7, 16, [16, 9, 3], 50), (23, 16, [5, 16], 1001),
                                          (40, 16, [13], 4133), (1, 16, [1, 1], 17),
                                          (39, 16, [16, 16, 16], 4096)])
def test_cin_stack_matches_einsum_conv1d(F0, D, units, B):
    """fx_cin_fwd/bwd through the autograd node vs the oracle's einsum + conv1d (fp32: sums of up
    to F0*Mi products, tolerance relative to the output scale)."""
    from fuxictr_amd import layers as L
    g = torch.Generator().manual_seed(F0 * D + B)
    x0 = torch.randn(B, F0, D, generator=g) * 0.5
    state, wb, prev = {}, [], F0
    for i, u in enumerate(units):
        W = torch.randn(u, F0 * prev, 1, generator=g) * (1.0 / (F0 * prev) ** 0.5)
        b = torch.randn(u, generator=g) * 0.1
        state["cin.cin_layer.layer_%d.weight" % (i + 1)] = W
        state["cin.cin_layer.layer_%d.bias" % (i + 1)] = b
        wb += [W, b]
        prev = u
    state["cin.fc.weight"] = torch.randn(1, sum(units), generator=g)
    state["cin.fc.bias"] = torch.zeros(1)
    leaves = [x0.clone().requires_grad_(True)] + [t.clone().requires_grad_(True) for t in wb]
    st = dict(state)
    for i in range(len(units)):
        st["cin.cin_layer.layer_%d.weight" % (i + 1)] = leaves[1 + 2 * i]
        st["cin.cin_layer.layer_%d.bias" % (i + 1)] = leaves[2 + 2 * i]
    ref = O.cin(st, "cin.", leaves[0], len(units))
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy)

    dl = [_dev(t).requires_grad_(True) for t in [x0] + wb]
    pooled = L._CINFn.apply(*dl)
    out = torch.nn.functional.linear(pooled, _dev(state["cin.fc.weight"]), _dev(state["cin.fc.bias"]))
    out.backward(_dev(gy))
    assert (out.detach().cpu() - ref.detach()).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    for a, r in zip(dl, leaves):
        scale = max(1.0, r.grad.abs().max().item())
        assert (a.grad.cpu() - r.grad).abs().max().item() <= 5e-5 * scale, (tuple(r.shape), scale)


@pytest.mark.gpu
@pytest.mark.parametrize("F0,Mi,O,B", [(39, 39, 16, 333), (39, 16, 16, 4096), (11, 20, 7, 65)])
def test_cin_packed_weight_image_is_only_a_layout(F0, Mi, O, B):
    """fx_cin_pack_w's image vs the kernels gathering W themselves (w_img = NULL): identical bits, in
    the forward, dX0 (accumulating), dXi and the dW partials."""
    import os
    if os.environ.get("FX_CIN_MFMA") == "0":
        pytest.skip("FX_CIN_MFMA=0: the fp32 VALU kernels take every shape, there is no image")
    g = torch.Generator().manual_seed(F0 + Mi + O)
    x0 = _dev(torch.randn(B, F0, 16, generator=g))
    xi = _dev(torch.randn(B, Mi, 16, generator=g))
    W = _dev(torch.randn(O, F0 * Mi, generator=g))
    bias = _dev(torch.randn(O, generator=g))
    gxn = _dev(torch.randn(B, O, 16, generator=g))
    gpool = _dev(torch.randn(B, O, generator=g))
    n = ops.cin_wimg_floats(F0, Mi, 16, O)
    assert n == F0 * (4 if Mi <= 16 else 10) * 64 + F0 * (1 if Mi <= 16 else 3) * 256
    assert ops.cin_wimg_floats(F0, Mi, 8, O) == 0 and ops.cin_wimg_floats(F0, Mi, 16, 17) == 0
    img = torch.empty(n, device=DEV)
    ops.cin_pack_w([(W, F0, Mi, img)], 16)
    outs = []
    for w_img in (img, None):
        xn = torch.empty(B, O, 16, device=DEV)
        pool = torch.empty(B, O, device=DEV)
        ops.cin_fwd(x0, xi, W, bias, xn, pool, w_img)
        dx0 = torch.ones(B, F0, 16, device=DEV)
        dxi = torch.empty(B, Mi, 16, device=DEV)
        partial = torch.empty(ops.cin_workgroups(), O * F0 * Mi + O, device=DEV)
        ops.cin_bwd(x0, xi, W, gxn, gpool, dx0, True, dxi, partial, w_img)
        outs.append([t.cpu() for t in (xn, pool, dx0, dxi, partial)])
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    had = torch.einsum("bhd,bmd->bhmd", x0.cpu().double(), xi.cpu().double()).reshape(B, -1, 16)
    ref = torch.einsum("oc,bcd->bod", W.cpu().double(), had) + bias.cpu().double().view(1, -1, 1)
    assert (outs[0][0].double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("D,adam", [(16, True), (1, True), (10, False), (16, False)])
def test_embedding_regularizer_dense_step_equals_dense_optimizer(D, adam):
    """fx_reg_stats / fx_reg_cross / sparse update with r(p) / fx_reg_dense_update together ==
    one dense Adam|SGD step on grad = scatter(G) + l1 sign(p) + l2 p with the global-norm clip."""
    R, l1, l2, lr = 3001, 3e-3, 2e-2, 0.05
    g = torch.Generator().manual_seed(D + 7 * adam)
    table0 = torch.randn(R, D, generator=g)
    table0[0].zero_()                                   # a padding row stays exactly zero
    table = _dev(table0)
    m = torch.zeros(R, D, device=DEV)
    v = torch.zeros(R, D, device=DEV)
    last = torch.zeros(R, dtype=torch.int32, device=DEV)
    scal = ops.new_scalars(DEV, lr=lr, max_norm=0.5)
    scal[_lib.SC_REG_L1:_lib.SC_REG_L1 + 1].fill_(l1)
    scal[_lib.SC_REG_L2:_lib.SC_REG_L2 + 1].fill_(l2)
    p_ref, m_ref, v_ref = table0.clone(), torch.zeros(R, D), torch.zeros(R, D)
    ws = torch.empty(ops.dedup_workspace_bytes(64), dtype=torch.uint8, device=DEV)
    parts = torch.zeros(3 * _lib.FX_REG_BLOCKS, device=DEV)
    cross = torch.zeros(_lib.FX_REG_CROSS_BLOCKS, device=DEV)
    for step in range(1, 4):
        ops.opt_begin_step(scal)
        ids = torch.randint(1, R, (64, 1), generator=g).numpy()
        dd = ops.dedup(_dev(ids, torch.int32), _dev([0], torch.int64), _dev([R], torch.int32),
                       _dev([-1], torch.int32), R, ws)
        nu = int(dd.n_unique.item())
        rows = dd.uniq_row[:nu].long().cpu()
        G = torch.zeros(64, D)
        G[:nu] = torch.randn(nu, D, generator=g) * 0.01
        ops.reg_stats(table, scal, parts)
        ops.reg_cross(table, D, dd, _dev(G), scal, cross)
        nb = _lib.FX_REG_BLOCKS
        r = l1 * torch.sign(p_ref) + l2 * p_ref
        np.testing.assert_allclose(parts[:nb].sum().item(), (p_ref.double() ** 2).sum().item(), rtol=1e-5)
        np.testing.assert_allclose(parts[nb:2 * nb].sum().item(), p_ref.double().abs().sum().item(), rtol=1e-5)
        np.testing.assert_allclose(parts[2 * nb:].sum().item(), (r.double() ** 2).sum().item(), rtol=1e-5)
        grad = r.clone()
        grad[rows] += G[:nu]
        sq = _dev([(G.double() ** 2).sum().item()], torch.float32)
        ops.clip_coef([sq, parts[2 * nb:], cross], scal)
        total = grad.double().norm().item()
        np.testing.assert_allclose(float(scal[_lib.SC_TOTAL_NORM]), total, rtol=2e-5)
        coef = min(1.0, 0.5 / (total + 1e-6))
        if adam:
            ops.sparse_adam(table, m, v, last, D, dd, _dev(G), scal)
            O.adam_dense(p_ref, grad * coef, m_ref, v_ref, step, lr)
        else:
            ops.sparse_sgd(table, D, dd, _dev(G), scal, last_step=last)
            p_ref -= lr * coef * grad
        ops.reg_dense_update(table, m if adam else None, v if adam else None, last, D, adam, scal)
        assert (table.cpu() - p_ref).abs().max().item() <= 2e-6
        assert table[0].abs().max().item() == 0.0
    if adam:
        assert (m.cpu() - m_ref).abs().max().item() <= 1e-6
        assert (v.cpu() - v_ref).abs().max().item() <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("n,ties", [(100, False), (4096, True), (300000, True), (65537, False),
                                     (2100000, False)])
def test_binary_metrics_match_sklearn(n, ties):
    """fx_binary_metrics == sklearn log_loss / roc_auc_score on the float64 view of the same float32
    predictions (what BaseModel.evaluate feeds them, rank_model.py:369-381), incl. heavy ties and
    saturated probabilities (exactly 0.0 / 1.0 -> the eps clip)."""
    from sklearn.metrics import log_loss, roc_auc_score
    rng = np.random.default_rng(n)
    p = rng.random(n).astype(np.float32)
    if ties:
        p = np.round(p * 50).astype(np.float32) / 50           # ~51 distinct values, 0.0 and 1.0 too
    y = (rng.random(n) < 0.2 + 0.6 * p).astype(np.float32)
    ll, auc = ops.binary_metrics(_dev(p), _dev(y))
    ref_ll = log_loss(y.astype(np.float64), p.astype(np.float64))
    ref_auc = roc_auc_score(y.astype(np.float64), p.astype(np.float64))
    assert abs(ll - ref_ll) <= 1e-12 * max(1.0, abs(ref_ll)), (ll, ref_ll)
    assert abs(auc - ref_auc) <= 1e-12, (auc, ref_auc)


@pytest.mark.gpu
def test_binary_metrics_one_class_raises_like_sklearn():
    with pytest.raises(ValueError):
        ops.binary_metrics(_dev(np.linspace(0.1, 0.9, 64).astype(np.float32)),
                           _dev(np.ones(64, dtype=np.float32)))


@pytest.mark.gpu
@pytest.mark.parametrize("N,B,fast", [(2, 4096, True), (8, 4096, True), (3, 777, True), (8, 512, False)])
def test_shard_plan_from_row_sorted_keys(N, B, fast):
    """fx_dedup (column fast path or generic) + fx_shard_plan(global_keys): every valid lookup's slot
    lies in its owner's bucket and carries the row's local index; padding lookups get the pad slot;
    buckets are filled in ascending row order and padded with the owner's pad row."""
    rng = np.random.default_rng(N * B)
    vocab = [50, 3, 7000, 911, 12]
    C = len(vocab)
    ids = np.stack([rng.integers(0, v, B) for v in vocab], 1).astype(np.int32)   # id 0 = padding
    if not fast:
        vocab = vocab[:3] + vocab[:2]            # repeated tables: columns not sorted -> generic path
        ids = np.stack([rng.integers(0, v, B) for v in vocab], 1).astype(np.int32)
        base = np.array([0, 50, 53, 0, 50], dtype=np.int64)
        total = 50 + 3 + 7000
    else:
        base = np.concatenate([[0], np.cumsum(vocab)[:-1]]).astype(np.int64)
        total = int(sum(vocab))
    pad = np.zeros(C, dtype=np.int32)
    n = B * C
    ws = torch.empty(ops.dedup_workspace_bytes(n), dtype=torch.uint8, device=DEV)
    dd = ops.dedup(_dev(ids), _dev(base), _dev(np.array(vocab, np.int32)), _dev(pad), total, ws,
                   want_uid=True, columns_sorted=fast)
    cap = int(np.ceil(1.5 * n / N / 64) * 64 + 64)
    send_idx = torch.empty(N * cap, dtype=torch.int32, device=DEV)
    uniq_slot = torch.empty(n, dtype=torch.int32, device=DEV)
    lookup_slot = torch.empty(B, C, dtype=torch.int32, device=DEV)
    scal = ops.new_scalars(DEV)
    pws = torch.empty(ops.shard_plan_workspace_ints(n, N), dtype=torch.int32, device=DEV)
    ops.shard_plan(dd, N, total, cap, send_idx, uniq_slot, lookup_slot, scal, global_keys=True,
                   workspace=pws)
    torch.cuda.synchronize()
    assert int(scal.view(torch.int32)[_lib.SC_ERR]) & _lib.FX_FLAG_A2A_OVERFLOW == 0
    send, ls = send_idx.cpu().numpy(), lookup_slot.cpu().numpy()
    rps = -(-total // N)
    g = ids.astype(np.int64) + base[None, :]
    valid = ids != 0
    assert (ls[~valid] == N * cap).all()
    owner = ls[valid] // cap
    np.testing.assert_array_equal(owner, g[valid] % N)
    np.testing.assert_array_equal(send[ls[valid]], g[valid] // N)
    for o in range(N):
        bucket = send[o * cap:(o + 1) * cap]
        k = int((bucket != rps).sum())
        assert (bucket[k:] == rps).all()
        rows = bucket[:k].astype(np.int64) * N + o
        assert (np.diff(rows) > 0).all()                       # ascending, unique
        want = np.unique(g[valid][g[valid] % N == o])
        got = rows if fast else rows      # the fast path also ships the tables' padding rows
        assert set(want).issubset(set(got.tolist()))
        assert len(got) <= len(want) + C
    # too small a capacity is flagged, not silently wrong
    scal2 = ops.new_scalars(DEV)
    ops.shard_plan(dd, N, total, 8, torch.empty(N * 8, dtype=torch.int32, device=DEV), uniq_slot,
                   lookup_slot, scal2, global_keys=True, workspace=pws)
    assert int(scal2.view(torch.int32)[_lib.SC_ERR]) & _lib.FX_FLAG_A2A_OVERFLOW


@pytest.mark.gpu
@pytest.mark.parametrize("D,B,lens", [(16, 4096, [50, 7]), (8, 333, [11]), (10, 100, [3, 64, 1]),
                                      (1, 257, [5]), (64, 64, [130])])
def test_seq_pooling_fused_into_the_gather_forward_and_backward(D, B, lens):
    """fx_emb_seq_pool_fwd + fx_emb_grad_reduce_scaled == F.embedding -> MaskedSum/AveragePooling
    (pooling.py:32-47, :59-70) and its autograd.  Forward tolerance 1e-6 x magnitude (fp32 sums of
    <= 130 rows in a different order), mask/denominator bit-exact."""
    g = torch.Generator().manual_seed(D * 1000 + B)
    n_seq = len(lens)
    vocabs = [37, 1000, 11][:n_seq] + [29]             # one table per sequence + one plain column
    bases = np.concatenate([[0], np.cumsum(vocabs)[:-1]]).tolist()
    R = int(sum(vocabs))
    table = torch.randn(R, D, generator=g)
    for b0 in bases:
        table[b0].zero_()                               # padding_idx 0 of every table
    modes = [ops.POOL_MEAN, ops.POOL_SUM, ops.POOL_MEAN][:n_seq]
    # columns: [plain | seq 0 | seq 1 ...]; slots: seq s -> slot s, plain -> last slot
    ids_cols, col_base, col_vocab, col_pad, col_off, col_den = [], [], [], [], [], []
    plain = torch.randint(0, vocabs[-1], (B, 1), generator=g)
    ids_cols.append(plain)
    col_base.append(bases[-1]); col_vocab.append(vocabs[-1]); col_pad.append(0)
    col_off.append(n_seq * D); col_den.append(-1)
    seq_col0, seq_ids = [], []
    for s, L in enumerate(lens):
        x = torch.randint(1, vocabs[s], (B, L), generator=g)
        n_valid = torch.randint(0, L + 1, (B,), generator=g)       # post-padded, some empty
        x[torch.arange(L).view(1, -1) >= n_valid.view(-1, 1)] = 0
        seq_col0.append(sum(c.shape[1] for c in ids_cols))
        ids_cols.append(x)
        seq_ids.append(x)
        col_base += [bases[s]] * L; col_vocab += [vocabs[s]] * L; col_pad += [0] * L
        col_off += [s * D] * L
        col_den += [s if modes[s] == ops.POOL_MEAN else -1] * L
    ids = torch.cat(ids_cols, dim=1).int()
    C = ids.shape[1]
    n_slots = n_seq + 1
    scal = ops.new_scalars(DEV)
    out = torch.full((B, n_slots * D), 9.0, device=DEV)
    denom = torch.empty(B, n_seq, device=DEV)
    d_tab, d_ids = _dev(table), _dev(ids)
    d_base, d_vocab = _dev(col_base, torch.int64), _dev(col_vocab, torch.int32)
    d_off = _dev(col_off, torch.int64)
    ops.emb_gather_fwd(d_tab, D, d_ids, d_base, d_vocab, d_off, None, None, None, out, scal,
                       n_cols=1)
    ops.emb_seq_pool_fwd(d_tab, D, d_ids, d_base, d_vocab, _dev(seq_col0, torch.int32),
                         _dev(lens, torch.int32), _dev(modes, torch.int32),
                         _dev([s * D for s in range(n_seq)], torch.int64), out, denom, scal)
    assert int(scal.view(torch.int32)[_lib.SC_ERR]) == 0
    # the reference formulation, differentiable w.r.t. the table
    W = table.clone().requires_grad_(True)
    pieces = []
    for s, L in enumerate(lens):
        e = torch.nn.functional.embedding(seq_ids[s] + bases[s], W)
        pieces.append(O.masked_average_pooling(e) if modes[s] == ops.POOL_MEAN
                      else O.masked_sum_pooling(e))
        mask_cnt = (e.sum(-1) != 0).float().sum(-1)
        assert torch.equal(denom[:, s].cpu(), mask_cnt + 1e-12)
    pieces.append(torch.nn.functional.embedding(plain[:, 0] + bases[-1], W))
    ref = torch.stack(pieces, dim=1)
    got = out.view(B, n_slots, D).cpu()
    assert torch.equal(got[:, n_seq], ref[:, n_seq].detach())
    tol = 1e-6 * max(1.0, float(ref.abs().max()))
    assert float((got - ref.detach()).abs().max()) <= tol
    # backward: d table through the pooled slots
    dout = torch.randn(B, n_slots * D, generator=g)
    ref.backward(dout.view(B, n_slots, D))
    ws = torch.empty(ops.dedup_workspace_bytes(B * C), dtype=torch.uint8, device=DEV)
    dd = ops.dedup(d_ids, d_base, d_vocab, _dev(col_pad, torch.int32), R, ws)
    G = torch.zeros(dd.n_max, D, device=DEV)
    sq = torch.empty(ops.emb_grad_reduce_partials(dd.n_max, D), device=DEV)
    scr = torch.zeros(ops.emb_grad_reduce_scratch_ints(dd.n_max), dtype=torch.int32, device=DEV)
    has_mean = any(j >= 0 for j in col_den)
    ops.emb_grad_reduce(_dev(dout), n_slots * D, d_off, C, D, dd, G, sq, scr,
                        _dev(col_den, torch.int32) if has_mean else None,
                        denom if has_mean else None)
    nu = int(dd.n_unique.item())
    rows = dd.uniq_row[:nu].cpu().long()
    gref = W.grad.clone()
    for b0 in bases:
        gref[b0].zero_()           # padding rows: masked by nn.Embedding(padding_idx), never sent
    touched = torch.zeros(R, dtype=torch.bool)
    touched[rows] = True
    assert not (~touched).any() or float(gref[~touched].abs().max()) == 0.0
    err = float((G[:nu].cpu() - gref[rows]).abs().max())
    assert err <= 2e-6 * max(1.0, float(gref.abs().max())), err


@pytest.mark.gpu
@pytest.mark.parametrize("R,L,vocab", [(8, 19968, 4220326), (2, 5000, 70001), (1, 777, 50),
                                       (3, 1, 10), (8, 300, 40)])
def test_dedup_sorted_runs_equals_the_generic_sort_path(R, L, vocab):
    """fx_dedup_sorted_runs (merge by rank counting) == fx_dedup (stable rocPRIM sort) on the same
    keys, every output array bit for bit: R ascending runs with duplicates ACROSS runs, pad ids at
    the tails, empty and full runs."""
    rng = np.random.default_rng(R * 1000 + L)
    pad = vocab - 1                                   # the shard's pad row
    runs = []
    for r in range(R):
        n_valid = [0, L][r] if r < 2 and L > 1 else int(rng.integers(0, L + 1))
        n_valid = min(n_valid, vocab - 1)
        vals = np.sort(rng.choice(vocab - 1, size=n_valid, replace=False)) if n_valid else \
            np.zeros(0, dtype=np.int64)
        runs.append(np.concatenate([vals, np.full(L - n_valid, pad)]))
    ids = _dev(np.concatenate(runs), torch.int32)
    n = R * L
    ws = torch.empty(ops.dedup_workspace_bytes(n), dtype=torch.uint8, device=DEV)
    a = ops.dedup_sorted_runs(ids, R, vocab, pad, ws)
    b = ops.dedup(ids.view(-1, 1), _dev([0], torch.int64), _dev([vocab], torch.int32),
                  _dev([pad], torch.int32), vocab, ws)
    nu = int(b.n_unique.item())
    assert int(a.n_unique.item()) == nu
    assert torch.equal(a.sorted_key, b.sorted_key) and torch.equal(a.sorted_pos, b.sorted_pos)
    assert torch.equal(a.uniq_row[:nu], b.uniq_row[:nu])
    assert torch.equal(a.seg_start[:nu + 1], b.seg_start[:nu + 1])
    ref = np.unique(np.concatenate([r_[r_ != pad] for r_ in runs]))
    assert np.array_equal(a.uniq_row[:nu].cpu().numpy().astype(np.int64), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("B,C,vocab,where", [
    (4096, 64, 2800000, "stream"), (4096, 64, 2800000, "graph"), (2049, 3, 1 << 24, "stream"),
    (8192, 26, 70000, "graph"), (300, 7, 200, "graph"), (32768, 26, 40000000, "stream")])
def test_dedup_generic_sort_on_side_stream_and_in_graph(B, C, vocab, where):
    """The generic path (one table shared by every column: c4's click_sequence + adgroup_id) is
    fx_sort.hip's LSD radix sort.  It must give numpy's answer — unique rows ascending, runs in
    ascending lookup position (stable) — on a non-null stream and when replayed from a captured
    hipGraph with fresh ids, which is how the training step runs it."""
    rng = np.random.default_rng(B * 31 + C)
    vocabs, bases = [vocab] * C, np.zeros(C, dtype=np.int64)
    ids_dev = torch.zeros(B, C, dtype=torch.int32, device=DEV)
    bases_d, vocab_d = _dev(bases, torch.int64), _dev(vocabs, torch.int32)
    pad_d = _dev([0] * C, torch.int32)
    ws = torch.empty(ops.dedup_workspace_bytes(B * C), dtype=torch.uint8, device=DEV)
    torch.cuda.synchronize()

    def draw():
        ids = np.minimum((vocab * rng.random((B, C)) ** 3).astype(np.int64), vocab - 1)
        ids[rng.random(ids.shape) < 0.2] = 0
        return ids

    def run():
        return ops.dedup(ids_dev, bases_d, vocab_d, pad_d, vocab, ws, want_uid=True)

    def check(dd, ids):
        keys = ids.reshape(-1)
        order = np.argsort(keys, kind="stable")
        order = order[keys[order] != 0]
        uniq, counts = np.unique(keys[order], return_counts=True)
        nu = int(dd.n_unique.item())
        assert nu == len(uniq)
        assert np.array_equal(dd.uniq_row[:nu].cpu().numpy().astype(np.int64) & 0xFFFFFFFF, uniq)
        seg = dd.seg_start[:nu + 1].cpu().numpy().astype(np.int64)
        assert np.array_equal(np.diff(seg), counts) and seg[0] == 0
        n_valid = len(order)
        pos = dd.sorted_pos[:n_valid].cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        assert np.array_equal(pos, order)                      # the whole stable permutation
        uid = dd.sorted_uid.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        assert np.array_equal(uid[:n_valid], np.repeat(np.arange(nu), counts))
        assert np.all(uid[n_valid:] == 0xFFFFFFFF)

    side = torch.cuda.Stream()
    if where == "stream":
        for _ in range(3):
            ids = draw()
            with torch.cuda.stream(side):
                ids_dev.copy_(torch.from_numpy(ids.astype(np.int32)))
                dd = run()
            side.synchronize()
            check(dd, ids)
        return
    with torch.cuda.stream(side):
        ids_dev.copy_(torch.from_numpy(draw().astype(np.int32)))
        run()
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        dd = run()
    for _ in range(4):
        ids = draw()
        ids_dev.copy_(torch.from_numpy(ids.astype(np.int32)))
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        check(dd, ids)


def test_catchup_replay_early_exit_matches_the_step_by_step_reference():
    """fx_adam_replay (round 4): rows leave the replay once a window of steps moved none of their elements.
    Against torch's own fp32 Adam stepped k times with zero gradients, over the magnitudes where the exit
    matters: p from 1e-4 (embedding init) to 50, m from 1e-10 to 0.1, sqrt(v) from 0.05 |m| to 30 |m|, rows
    10 ... 300 steps behind (300 > FX_REPLAY_MAX).  Bound: 1.2e-5 of the row's total movement + 2 ulp of p —
    the replay's arithmetic, not the exit: rcp / rsq / the running sqrt, and for rows replayed from the first
    steps of a run 1 - beta2^t formed in fp32 (torch forms it in double: 6e-5 relative at t = 1; measured
    worst 8e-6 of the movement for a row replayed from t = 0)."""
    gen = torch.Generator().manual_seed(3)
    D, T, lr = 16, 300, 1e-3
    lasts = [0, 45, 100, 200, 260, 290]
    rows = []
    for pm in (1e-4, 1e-2, 1.0, 50.0):
        for mm in (1e-10, 1e-6, 1e-3, 1e-1):
            for c in (0.05, 1.0, 30.0):
                for last in lasts:
                    rows.append((pm, mm, c, last))
    R = len(rows)
    p0 = torch.empty(R, D)
    m0 = torch.empty(R, D)
    v0 = torch.empty(R, D)
    for i, (pm, mm, c, _) in enumerate(rows):
        p0[i] = pm * (0.5 + torch.rand(D, generator=gen)) * torch.sign(torch.randn(D, generator=gen))
        m0[i] = mm * torch.randn(D, generator=gen)
        v0[i] = (c * mm * (0.5 + torch.rand(D, generator=gen))) ** 2
    last0 = torch.tensor([r[3] for r in rows], dtype=torch.int32)
    table, m, v, last = _dev(p0), _dev(m0), _dev(v0), last0.to(DEV)
    scal = ops.new_scalars(DEV, lr=lr)
    for _ in range(T):
        ops.opt_begin_step(scal)
    ops.adam_catchup(table, m, v, last, D, None, R, 0, scal)
    assert int(last.min()) == T and int(last.max()) == T
    p_ref, m_ref, v_ref = p0.clone(), m0.clone(), v0.clone()
    zero = torch.zeros(1, D)
    for i, (_, _, _, l0) in enumerate(rows):
        pi, mi, vi = p_ref[i:i + 1], m_ref[i:i + 1], v_ref[i:i + 1]
        for t in range(l0 + 1, T + 1):
            _adam_ref_step(pi, zero, mi, vi, t, lr)
    moved = (p_ref - p0).abs().max(dim=1, keepdim=True).values
    ulp = p_ref.abs() * 2.0 ** -23
    err = (table.cpu() - p_ref).abs()
    bound = 1.2e-5 * moved + 2 * ulp + 1e-30
    worst = (err / bound).max().item()
    assert worst <= 1.0, (worst, rows[int((err / bound).max(dim=1).values.argmax())])
    # v: closed form; m: sequential inside the replayed steps, closed form for the rest
    assert ((v.cpu() - v_ref).abs() <= 1e-4 * v_ref.abs() + 1e-38).all()
    assert ((m.cpu() - m_ref).abs() <= 1e-4 * m_ref.abs() + 1e-38).all()
