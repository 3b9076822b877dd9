"""The Adam series table (round 6; include/fxctr.h fx_adam_series_build, csrc/fx_series.hip, fx_series_move in
csrc/fx_common.h) on a real MI355X.

What is pinned here, against fp64 restatements of the reference's arithmetic (torch.optim.Adam with a zero
gradient, torch_utils.py:72-76 as stepped at rank_model.py:322):
  * every table entry F(t; c) against the directly summed series, c from 0 (v >> eps^2) to far past the
    eps-dominated regime (v ~ eps^2 and below);
  * the catch-up kernels with the table behind the scalar block against the k zero-gradient Adam steps done
    one by one in fp64 — the D = 16 + D = 1 pair (quad path), D = 16 alone, and the generic rows (D = 8, 10, 1) —
    for gaps 13 ... 5000 at steps 1 ... 60 000, moments over twelve orders of magnitude, never-touched rows,
    v = 0 elements;
  * the short gaps (<= FX_SERIES_KDIR) still take the step-by-step replay and meet the same bound.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from fuxictr_amd import _lib, ops  # noqa: E402

DEV = "cuda:0"
B1, B2, EPS = 0.9, 0.999, 1e-8          # torch.optim.Adam's python-side doubles: the bias corrections
B1F, B2F = float(np.float32(B1)), float(np.float32(B2))     # what the fp32 tensor ops multiply the moments by
KDIR, EARLY = 12, 128


def _dev(x, dtype=None):
    t = torch.as_tensor(x)
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV).contiguous()


def _wg(t, n=512):
    i = np.arange(1, n + 1, dtype=np.float64)
    w = B1F ** i / (1 - B1 ** (t + i))
    g = B2F ** (i / 2) / np.sqrt(1 - B2 ** (t + i))
    return w, g


def _table(scal):
    """-> (tcap, early [128, 8, 8], main [tcap - 128, 8]) read back from the allocation behind `scal`."""
    tcap = int(scal.view(torch.int32)[_lib.SC_SERIES_TCAP].item())
    assert tcap > EARLY
    words = int(_lib.load().fx_adam_series_words(tcap))
    base = scal.untyped_storage()
    full = torch.empty(0, dtype=torch.float32, device=scal.device).set_(base, 0, (_lib.SC_WORDS + words,))
    body = full[_lib.SC_WORDS + 16:].cpu().numpy()
    early = body[:EARLY * 64].reshape(EARLY, 8, 8)
    main = body[EARLY * 64:].reshape(tcap - EARLY, 8)
    return tcap, early, main, full[_lib.SC_WORDS:_lib.SC_WORDS + 16].cpu()


def _eval(segs, c):
    tot = 0.0
    for s in segs:
        gs, c0, c2, c3, c4, c5, c6 = [float(x) for x in s[:7]]
        y = gs / (gs + c)
        tot += y * (c0 + y * y * (c2 + y * (c3 + y * (c4 + y * (c5 + y * c6)))))
    return tot


def test_series_table_entries_match_the_directly_summed_series():
    scal = ops.new_scalars(DEV, series=True)
    tcap, early, main, hdr = _table(scal)
    assert tcap == ops.series_tcap(B1, B2) and tcap >= 18014      # both bias corrections are 1 in fp32 from there
    assert int(hdr.view(torch.int32)[1]) == tcap
    err_hdr = float(hdr[2])
    assert 0.0 < err_hdr <= ops.SERIES_MAX_ERR and ops.series_error(scal) == pytest.approx(err_hdr)
    worst, nsegs = 0.0, {}
    ts = list(range(1, 140)) + [150, 200, 255, 256, 1000, 5000, tcap - 1]
    for t in ts:
        if t < EARLY:
            n = int(early[t, 0, 7:8].view(np.int32)[0])
            segs = early[t, :n]
            assert n in (1, 2, 4, 8) and not early[t, n:, :7].any()
        else:
            n, segs = 1, main[t - EARLY][None]
        nsegs[t] = n
        w, g = _wg(t)
        for r in (0.0, 1e-3, 0.05, 0.5, 2.0, 30.0, 1e3, 1e6):
            c = r * g[0]
            ex = float((w / (g + c)).sum())
            e = abs(_eval(segs, c) - ex) / ex
            worst = max(worst, e)
            assert e <= 3e-7, (t, r, e)
    # the ladder shortens as the bias corrections flatten: one segment from t = 128 on
    assert nsegs[1] == 8 and nsegs[64] <= 2 and nsegs[127] <= 2 and nsegs[1000] == 1
    assert worst <= 1.1 * err_hdr + 6e-8       # (the builder probes other values of c: same order)
    print("[series] tcap %d, worst entry error (builder) %.2e, (this probe) %.2e, segments at t=1/32/64/127: %s"
          % (tcap, err_hdr, worst, [nsegs[t] for t in (1, 32, 64, 127)]))


def test_series_is_left_out_when_asked_or_when_the_betas_do_not_converge(monkeypatch):
    assert int(ops.new_scalars(DEV).view(torch.int32)[_lib.SC_SERIES_TCAP].item()) == 0
    monkeypatch.setenv("FX_CATCHUP_SERIES", "0")
    assert int(ops.new_scalars(DEV, series=True).view(torch.int32)[_lib.SC_SERIES_TCAP].item()) == 0
    monkeypatch.delenv("FX_CATCHUP_SERIES")
    # beta1 / sqrt(beta2) = 0.9995: the sum has not converged after 512 terms — the builder says so, the host
    # keeps the replay
    s = ops.new_scalars(DEV, beta1=0.999, beta2=0.999, series=True)
    assert int(s.view(torch.int32)[_lib.SC_SERIES_TCAP].item()) == 0


def _exact_rows(p, m, v, last, upto, lr):
    """k = upto - last zero-gradient Adam steps of every row, one by one, in fp64."""
    p = p.astype(np.float64).copy()
    m = m.astype(np.float64).copy()
    v = v.astype(np.float64).copy()
    dp = np.zeros_like(p)
    for r in range(p.shape[0]):
        k = int(upto - last[r])
        if k <= 0:
            continue
        kk = min(k, 4000)                       # (0.9^4000: nothing left)
        i = np.arange(1, kk + 1, dtype=np.float64)[:, None]
        t = last[r] + i
        mi = m[r][None, :] * B1F ** i
        vi = v[r][None, :] * B2F ** i
        u = lr / (1 - B1 ** t) * mi / (np.sqrt(vi) / np.sqrt(1 - B2 ** t) + EPS)
        dp[r] = u.sum(0)
        m[r] *= B1F ** k
        v[r] *= B2F ** k
    return p - dp, m, v, dp


@pytest.mark.parametrize("dims", [(16, 1), (16,), (8,), (10, 1), (1,)])
@pytest.mark.parametrize("upto", [14, 60, 333, 2000, 60000])
def test_series_catchup_equals_zero_gradient_adam_stepped_k_times_in_fp64(dims, upto):
    rng = np.random.default_rng(upto * 31 + sum(dims))
    R = 1500
    lr = 1e-3
    scal = ops.new_scalars(DEV, lr=lr, series=True)
    scal.view(torch.int32)[_lib.SC_STEP] = upto
    ops.opt_begin_step(scal)                                   # step = upto + 1: rows are brought to `upto`
    # gaps: short (replayed), just past the switch, long, and beyond the table's 1024-step tail cut
    gaps = rng.choice([1, 2, 5, 12, 13, 14, 20, 37, 100, 255, 256, 257, 700, 1023, 1024, 1500, 5000], size=R)
    last = np.maximum(upto - gaps, 0).astype(np.int32)
    last[:8] = upto                                            # nothing to do
    host, states = [], []
    for D in dims:
        # gradients from 1e-9 (v ~ 1e-21: c = eps / sqrt(v) ~ 300, eps-dominated) to 1e-1
        gmag = 10.0 ** rng.uniform(-9, -1, size=(R, 1))
        p = rng.normal(size=(R, D)).astype(np.float32) * 1e-2
        m = (gmag * (1 - B1) * rng.uniform(0.2, 3, size=(R, D)) * rng.choice([-1, 1], size=(R, D))).astype(np.float32)
        v = (gmag ** 2 * (1 - B2) * rng.uniform(0.2, 30, size=(R, D))).astype(np.float32)
        m[8:40] = 0
        v[8:40] = 0                                            # rows that never had a gradient
        v[40:50, 0] = 0                                        # an element whose g^2 underflowed ...
        m[40:50, 0] = 1e-25                                    # ... beside a moment that did not
        m[50:60, -1] = 0                                       # an element at rest next to moving ones
        m[last == 0] = 0
        v[last == 0] = 0                                       # last_step = 0 <=> never updated
        host.append((p, m, v))
        states.append(ops.RowState(_dev(p), _dev(m), _dev(v), _dev(last.copy()), D))
    ids = np.arange(R, dtype=np.int64)[:, None]
    ws = torch.empty(ops.dedup_workspace_bytes(R), dtype=torch.uint8, device=DEV)
    dd = ops.dedup(_dev(ids, torch.int32), _dev([0], torch.int64), _dev([R], torch.int32), _dev([-1], torch.int32),
                   R, ws)
    ops.adam_catchup_rows(states, dd, -1, scal)
    torch.cuda.synchronize()
    worst = 0.0
    for (p, m, v), st in zip(host, states):
        pe, me, ve, dp = _exact_rows(p, m, v, last, upto, lr)
        pg = st.table.cpu().numpy().astype(np.float64)
        live = (m != 0) & (v > 0)
        # rows / elements that cannot move stay bit-identical
        assert np.array_equal(st.table.cpu().numpy()[~live], p[~live])
        # the move itself to 1.5e-6 of its size (the table entries carry <= 2.5e-7, fp32 evaluation the rest)
        # + one rounding of p
        err = np.abs((p.astype(np.float64) - pg) - dp)
        bound = 1.5e-6 * np.abs(dp) + 6.0e-8 * np.abs(p) + 1e-12
        ratio = (err / bound).max(1)
        by_gap = {int(gp): round(float(ratio[gaps == gp].max()), 2) for gp in np.unique(gaps)}
        assert ratio.max() <= 1.0, (dims, st.D, upto, by_gap)
        worst = max(worst, float(ratio.max()))
        assert np.allclose(st.m.cpu().numpy(), me, rtol=3e-7, atol=1e-38)
        assert np.allclose(st.v.cpu().numpy(), ve, rtol=3e-7, atol=1e-38)
        assert np.array_equal(st.last_step.cpu().numpy(), np.maximum(last, upto))
    print("[series catch-up] dims %s upto %d: worst error / bound = %.2f" % (dims, upto, worst))
