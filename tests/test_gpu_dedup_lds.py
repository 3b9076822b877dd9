"""The bucketed in-LDS de-dup (csrc/fx_dedup_lds.hip, fx_dedup with columns_sorted = 2) on a real MI355X: every
output array bit for bit against a numpy restatement of its contract —
    valid lookups (id in range, != padding_idx) stably sorted by (row & 255, row >> 8, position);
    uniq_row / seg_start / n_unique / sorted_uid over that order; the padding tail (key = total_rows,
    position = unique index = 0xFFFFFFFF)
— and, order aside, the same unique rows and per-row position lists as the generic (ascending) path.
Edge cases the domain has: c4's schema (a 50-position sequence that aliases its target's table, post-padded with
0, beside 14 one-column tables, three of them with 2 - 3 rows), every lookup on ONE row (a bucket of 209 K pairs:
the global-memory form of the bucket sort), two hot rows in one bucket, all lookups padding, one lookup,
vocabularies of <= 256 rows (no LDS pass at all) and of 2^25 rows (three), B * C that is not a multiple of
anything, a non-null stream and hipGraph replay with fresh ids (how the training step runs it)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from fuxictr_amd import ops  # noqa: E402
from test_gpu_fused import DEV, _dev  # noqa: E402


def _expect(ids, bases, vocabs, pads, total):
    B, C = ids.shape
    keys = ids.astype(np.int64) + np.asarray(bases)[None, :]
    valid = (ids >= 0) & (ids < np.asarray(vocabs)[None, :]) & (ids != np.asarray(pads)[None, :])
    pos = np.arange(B * C, dtype=np.int64)
    k, p = keys.reshape(-1)[valid.reshape(-1)], pos[valid.reshape(-1)]
    order = np.lexsort((p, k >> 8, k & 255))
    k, p = k[order], p[order]
    n = B * C
    nv = len(k)
    skey = np.full(n, total, dtype=np.int64)
    spos = np.full(n, 0xFFFFFFFF, dtype=np.int64)
    suid = np.full(n, 0xFFFFFFFF, dtype=np.int64)
    skey[:nv], spos[:nv] = k, p
    head = np.ones(nv, dtype=bool)
    head[1:] = k[1:] != k[:-1]
    suid[:nv] = np.cumsum(head) - 1
    uniq = k[head]
    seg = np.concatenate([np.nonzero(head)[0], [nv]])
    return skey, spos, suid, uniq, seg


def _run(ids, bases, vocabs, pads, total, grouped=True, ws=None):
    B, C = ids.shape
    if ws is None:
        ws = torch.empty(ops.dedup_workspace_bytes(B * C), dtype=torch.uint8, device=DEV)
    return ops.dedup(_dev(ids, torch.int32), _dev(bases, torch.int64), _dev(vocabs, torch.int32),
                     _dev(pads, torch.int32), total, ws, want_uid=True, grouped=grouped)


def _check(dd, exp):
    skey, spos, suid, uniq, seg = exp
    nu = int(dd.n_unique.item())
    assert nu == len(uniq)
    u32 = lambda t: t.cpu().numpy().astype(np.int64) & 0xFFFFFFFF   # noqa: E731
    assert np.array_equal(u32(dd.sorted_key), skey)
    assert np.array_equal(u32(dd.sorted_pos), spos)
    assert np.array_equal(u32(dd.sorted_uid), suid)
    assert np.array_equal(u32(dd.uniq_row[:nu]), uniq)
    assert np.array_equal(u32(dd.seg_start[:nu + 1]), seg)


def _schema(vocabs):
    bases = np.concatenate([[0], np.cumsum(vocabs)[:-1]]).astype(np.int64)
    return bases, int(sum(vocabs))


def _taobao_ids(rng, B, L=50, scale=1.0, dist="powerlaw"):
    """c4's id plan: 14 one-column tables, then L positions of a sequence that aliases table 1 (adgroup_id)."""
    cards = [1141729, 846811, 2, 6769, 423436, 255875, 99815, 97, 13, 2, 7, 4, 3, 2]
    vocabs = [max(2, int(c * scale)) + 1 for c in cards]
    bases, total = _schema(vocabs)
    cols, cb, cv = [], [], []
    for c, v in enumerate(vocabs):
        u = rng.random(B)
        cols.append(np.minimum((np.floor((v - 1) * (u ** 3 if dist == "powerlaw" else u))).astype(np.int64) + 1, v - 1))
        cb.append(bases[c])
        cv.append(v)
    v = vocabs[1]
    u = rng.random((B, L))
    seq = np.minimum(np.floor((v - 1) * (u ** 3 if dist == "powerlaw" else u)).astype(np.int64) + 1, v - 1)
    lens = rng.integers(1, L + 1, size=B)
    seq[np.arange(L)[None, :] >= lens[:, None]] = 0
    ids = np.concatenate([np.stack(cols, axis=1), seq], axis=1)
    cb += [bases[1]] * L
    cv += [v] * L
    return ids, np.asarray(cb, np.int64), cv, [0] * len(cv), total


@pytest.mark.parametrize("dist", ["powerlaw", "uniform"])
@pytest.mark.parametrize("B,scale", [(4096, 1.0), (4096, 0.01), (1000, 1.0), (10000, 0.1)])
def test_c4_schema_bit_exact(B, scale, dist):
    rng = np.random.default_rng(B + int(scale * 100))
    ids, bases, vocabs, pads, total = _taobao_ids(rng, B, scale=scale, dist=dist)
    dd = _run(ids, bases, vocabs, pads, total)
    torch.cuda.synchronize()
    _check(dd, _expect(ids, bases, vocabs, pads, total))


@pytest.mark.parametrize("case", ["one_row", "two_hot_same_bucket", "all_pad", "single", "tiny_vocab", "big_vocab",
                                  "ragged", "bad_ids", "hot_plus_tail"])
def test_edge_cases_bit_exact(case):
    rng = np.random.default_rng(len(case))
    pads = None
    if case == "one_row":                       # 209 K lookups of one row: the bucket does not fit LDS
        B, C, V = 4096, 51, 900000
        ids = np.full((B, C), 123457)
    elif case == "two_hot_same_bucket":         # rows 5 and 5 + 256 * 77 share bucket 5; > 8192 pairs together
        B, C, V = 4096, 8, 1 << 20
        ids = rng.integers(1, V, (B, C))
        hot = rng.random((B, C))
        ids[hot < 0.3] = 5
        ids[(hot >= 0.3) & (hot < 0.6)] = 5 + 256 * 77
    elif case == "hot_plus_tail":               # an oversized bucket whose pairs need all three global passes
        B, C, V = 8192, 6, (1 << 25) + 3
        ids = rng.integers(1, V, (B, C))
        m = rng.random((B, C)) < 0.4
        ids[m] = (rng.integers(0, 1 << 17, int(m.sum())) << 8) | 9      # bucket 9: ~ 19 K pairs, many rows
    elif case == "all_pad":
        B, C, V = 777, 5, 1000
        ids = np.zeros((B, C), dtype=np.int64)
    elif case == "single":
        B, C, V = 1, 1, 10
        ids = np.array([[7]])
    elif case == "tiny_vocab":                  # <= 256 rows in all: every bucket is one row, no LDS pass
        B, C, V = 3000, 4, 60
        ids = rng.integers(0, V, (B, C))
    elif case == "big_vocab":                   # 26-bit rows: three LDS passes
        B, C, V = 4096, 26, 1 << 25
        ids = np.minimum((V * rng.random((B, C)) ** 3).astype(np.int64) + 1, V - 1)
    elif case == "ragged":
        B, C, V = 8191, 3, 70001
        ids = rng.integers(0, V, (B, C))
    else:                                       # out-of-range ids are dropped like padding
        B, C, V = 2048, 4, 5000
        ids = rng.integers(-3, V + 3, (B, C))
    if case in ("tiny_vocab",):
        vocabs = [V] * C
        bases, total = _schema(vocabs)
    else:                                       # every column looks up the SAME table
        vocabs, bases, total = [V] * C, np.zeros(C, np.int64), V
    pads = [0] * C
    dd = _run(ids, bases, vocabs, pads, total)
    torch.cuda.synchronize()
    _check(dd, _expect(ids, bases, vocabs, pads, total))


def test_same_rows_and_position_lists_as_the_ascending_path():
    rng = np.random.default_rng(17)
    ids, bases, vocabs, pads, total = _taobao_ids(rng, 4096)
    a = _run(ids, bases, vocabs, pads, total, grouped=True)
    b = _run(ids, bases, vocabs, pads, total, grouped=False)
    nu = int(a.n_unique.item())
    assert nu == int(b.n_unique.item())
    ua, ub = a.uniq_row[:nu].cpu().numpy(), b.uniq_row[:nu].cpu().numpy()
    assert np.all(np.diff(ub.astype(np.int64)) > 0)               # the generic path: ascending
    perm = np.argsort(ua.astype(np.int64), kind="stable")
    assert np.array_equal(ua[perm], ub)
    sa, sb = a.seg_start[:nu + 1].cpu().numpy().astype(np.int64), b.seg_start[:nu + 1].cpu().numpy().astype(np.int64)
    pa, pb = a.sorted_pos.cpu().numpy(), b.sorted_pos.cpu().numpy()
    for j in list(range(0, nu, max(1, nu // 400))) + [nu - 1]:
        u = perm[j]
        assert np.array_equal(pa[sa[u]:sa[u + 1]], pb[sb[j]:sb[j + 1]])


@pytest.mark.parametrize("where", ["stream", "graph"])
def test_on_a_side_stream_and_under_graph_replay(where):
    rng = np.random.default_rng(23)
    B = 4096
    ids0, bases, vocabs, pads, total = _taobao_ids(rng, B)
    C = ids0.shape[1]
    ids_dev = torch.zeros(B, C, dtype=torch.int32, device=DEV)
    bd, vd, pd = _dev(bases, torch.int64), _dev(vocabs, torch.int32), _dev(pads, torch.int32)
    ws = torch.empty(ops.dedup_workspace_bytes(B * C), dtype=torch.uint8, device=DEV)
    torch.cuda.synchronize()

    def run():
        return ops.dedup(ids_dev, bd, vd, pd, total, ws, want_uid=True, grouped=True)

    side = torch.cuda.Stream()
    if where == "stream":
        for _ in range(3):
            ids = _taobao_ids(rng, B)[0]
            ids_dev.copy_(torch.from_numpy(ids).int())
            torch.cuda.synchronize()
            with torch.cuda.stream(side):
                dd = run()
            side.synchronize()
            _check(dd, _expect(ids, bases, vocabs, pads, total))
        return
    with torch.cuda.stream(side):
        run()
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            dd = run()
    torch.cuda.synchronize()
    for _ in range(3):
        ids = _taobao_ids(rng, B)[0]
        ids_dev.copy_(torch.from_numpy(ids).int())
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        _check(dd, _expect(ids, bases, vocabs, pads, total))
