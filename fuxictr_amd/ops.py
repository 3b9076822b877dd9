"""Thin tensor-level wrappers over the C-ABI (include/fxctr.h).

torch is plumbing here: it owns device memory and the stream; every computation below is a call
into libfxctr.so.  All functions require CUDA(HIP) fp32/int32 contiguous tensors and raise
otherwise — there is no CPU path.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr, vp

_DT = {torch.float32: _lib.FX_F32, torch.float64: _lib.FX_F64,
       torch.int32: _lib.FX_I32, torch.int64: _lib.FX_I64}


class KernelTimer(object):
    """Opt-in kernel timing for bench.py's roofline objects, off (zero cost) otherwise.

    Per-launch HIP events are useless at this kernel size on this platform: an event pair around
    one launch adds ~10 us (measured: 87.7 us for a GEMM rocprofv3 times at 75 us, 12.5 us for the
    8.5 us gather, even with the stream kept busy).  So the launches of eager training steps are
    RECORDED (entry point + its live arguments, tagged with the step they belong to) and then each
    recorded launch group is REPLAYED `reps` times back to back between one event pair:
    average = kernel duration + the ~1.3 us dependent-launch boundary — what rocprofv3's kernel
    trace reports, within a few per cent.

    HBM-bound groups (`multi_step`: the sparse path, the gather on its own) are replayed ROUND-ROBIN over
    ALL recorded steps — each step holds its own batch, ids, de-dup result and gradients — so that a
    launch meets rows it has not seen for (steps - 1) other batches, as inside the training step; a
    single batch replayed back to back would find its rows in the 256 MB Infinity Cache.
    MFMA-bound groups (one GEMM shape) use the first recorded step only.

    The replay RE-EXECUTES the recorded launches with their live arguments: the optimizer step counter
    advances, row updates are applied again, outputs are overwritten.  The model / optimizer state is
    no longer the trained trajectory afterwards — bench.py discards the model after this pass."""
    recording = False
    calls = []       # (name, group, work, fn, args, kwargs, step)
    step = 0
    multi_step = ("sparse_path",)

    @classmethod
    def note(cls, name, group, work, fn, args, kwargs):
        if cls.step > 0 and not ((group or name) in cls.multi_step or name.endswith("@alone")):
            return                     # MFMA-bound groups: the first recorded step is enough
        cls.calls.append((name, group, work, fn, args, kwargs, cls.step))

    @classmethod
    def next_step(cls):
        cls.step += 1

    @classmethod
    def replay(cls, reps=20):
        """-> name/group -> dict(launches, total_ms, avg_us, work) PER RECORDED STEP.
        The recorded launches of one group (a GEMM shape, the sparse path) are captured, in step
        order, into ONE hipGraph; the graph is replayed `reps` times between one event pair.  That
        is how the timed region itself executes them (graph launch, dependent boundaries, the same
        mix of shapes following each other), so the averages line up with rocprofv3's kernel trace
        of the timed region."""
        was, cls.recording = cls.recording, False
        groups = {}
        for c in cls.calls:
            groups.setdefault(c[1] or c[0], []).append(c)
        out = {}
        try:
            side = torch.cuda.Stream()
            for key, calls in groups.items():
                steps = sorted(set(c[6] for c in calls))
                if not (key in cls.multi_step or key.endswith("@alone")):
                    calls = [c for c in calls if c[6] == steps[0]]
                    steps = steps[:1]
                for c in calls:                                      # warm (allocations, caches)
                    c[3](*c[4], **c[5])
                torch.cuda.synchronize()
                # a group of ONE launch: ten copies of it in the graph, so that the graph launch itself
                # (several us) does not sit in a 10-us kernel's figure
                copies = 10 if len(calls) == 1 else 1
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    for _ in range(copies):
                        for c in calls:
                            c[3](*c[4], **c[5])
                g.replay()
                torch.cuda.synchronize()
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    g.replay()
                e1.record()
                torch.cuda.synchronize()
                per_call_ms = e0.elapsed_time(e1) / reps / (len(calls) * copies)
                if not key.endswith("@alone"):
                    # every launch of one step, whatever its group: bench.py's `kernel_sum_us`
                    tot = out.setdefault("__step__", {"launches": 0.0, "total_ms": 0.0, "work": 0.0,
                                                      "steps": 1})
                    tot["launches"] += len(calls) / float(len(steps))
                    tot["total_ms"] += per_call_ms * len(calls) / len(steps)
                for c in calls:
                    for k in (c[0], c[1]):
                        if k is None:
                            continue
                        o = out.setdefault(k, {"launches": 0.0, "total_ms": 0.0, "work": 0.0,
                                               "steps": len(steps)})
                        o["launches"] += 1.0 / len(steps)
                        o["total_ms"] += per_call_ms / len(steps)
                        o["work"] += float(c[2]) / len(steps)
                del g
        finally:
            cls.recording = was
        for o in out.values():
            o["avg_us"] = 1e3 * o["total_ms"] / max(o["launches"], 1e-9)
        return out

    @classmethod
    def reset(cls):
        cls.calls = []
        cls.step = 0


def _timed(name, group=None, work=None, alone=False):
    """Decorator: when KernelTimer is recording, note this C-ABI call (it still runs normally).
    alone: besides its group, time the call on its own under `name` (idempotent calls only)."""
    def deco(fn):
        def wrapper(*a, **kw):
            if KernelTimer.recording:
                w = work(*a, **kw) if work else 0
                KernelTimer.note(name, group, w, fn, a, kw)
                if alone and group:
                    KernelTimer.note(name + "@alone", None, w, fn, a, kw)
            return fn(*a, **kw)
        wrapper.__name__ = fn.__name__
        wrapper.__doc__ = fn.__doc__
        wrapper.__wrapped__ = fn
        return wrapper
    return deco


def _need_cuda(t, name):
    if not t.is_cuda:
        raise _lib.FxError("%s must live on the GPU: the native path has no CPU fallback "
                           "(got device %s)" % (name, t.device))


# worst relative error of a table entry (against the directly summed series) that the catch-up accepts: the
# replay it replaces carries ~2e-7 from its approximate reciprocals
SERIES_MAX_ERR = 2.5e-7
_series_cache = {}          # (device, beta1, beta2) -> (tcap, table words incl. header) of a checked table


def series_tcap(beta1, beta2):
    """Entries of the Adam series table: one per step until both bias corrections are 1 in fp32."""
    import math
    if not (0.0 < beta1 < 1.0 and 0.0 < beta2 < 1.0):
        return 0
    t = max(math.log(2.0 ** -26) / math.log(beta1), math.log(2.0 ** -26) / math.log(beta2))
    t = int(math.ceil(t / 1024.0)) * 1024
    return t if 256 <= t <= (1 << 20) else 0


def new_scalars(device, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, max_norm=0.0, series=False):
    """Allocate and initialise a device-resident fx_scalars block (16 x 4-byte words).  series=True (the
    exact-mode Adam of the tables): the block is followed, in the same allocation, by the Adam series table
    of (beta1, beta2) — fx_adam_series_build; FX_CATCHUP_SERIES=0 leaves it out (the step-by-step replay)."""
    host = torch.zeros(_lib.SC_WORDS, dtype=torch.float32)
    host[_lib.SC_LR] = lr
    host[_lib.SC_BETA1] = beta1
    host[_lib.SC_BETA2] = beta2
    host[_lib.SC_EPS] = eps
    host[_lib.SC_CLIP] = 1.0
    host[_lib.SC_MAX_NORM] = max_norm
    device = torch.device(device)
    tcap = series_tcap(beta1, beta2) if series and os.environ.get("FX_CATCHUP_SERIES", "1") != "0" else 0
    if tcap == 0 or device.type != "cuda":
        return host.to(device)
    lib = _lib.load()
    words = int(lib.fx_adam_series_words(tcap))
    buf = torch.zeros(_lib.SC_WORDS + words, dtype=torch.float32, device=device)
    scal = buf[:_lib.SC_WORDS]              # (a view: the table lives as long as the block)
    scal.copy_(host)
    key = (str(device), float(host[_lib.SC_BETA1]), float(host[_lib.SC_BETA2]))      # (as fp32 holds them)
    hit = _series_cache.get(key)
    if hit is None:
        check(lib.fx_adam_series_build(ptr(scal), tcap, stream_ptr(device)), "fx_adam_series_build")
        err = float(buf[_lib.SC_WORDS + 2].item())          # one host sync per (device, betas)
        ok = err <= SERIES_MAX_ERR
        _series_cache[key] = (tcap, buf[_lib.SC_WORDS:].clone() if ok else None, err)
        if not ok:
            import logging
            logging.warning("Adam series table for betas (%g, %g): worst entry error %.2e > %.1e - the "
                            "exact-mode catch-up keeps the step-by-step replay", beta1, beta2, err,
                            SERIES_MAX_ERR)
            scal.view(torch.int32)[_lib.SC_SERIES_TCAP:_lib.SC_SERIES_TCAP + 1].fill_(0)
    elif hit[1] is not None:
        buf[_lib.SC_WORDS:].copy_(hit[1])
        scal.view(torch.int32)[_lib.SC_SERIES_TCAP:_lib.SC_SERIES_TCAP + 1].fill_(tcap)
    return scal


def series_error(scal):
    """Worst entry error the builder measured for the table behind `scal` (None: no table)."""
    if int(scal.view(torch.int32)[_lib.SC_SERIES_TCAP].item()) == 0:
        return None
    hit = _series_cache.get((str(scal.device), float(scal[_lib.SC_BETA1].item()), float(scal[_lib.SC_BETA2].item())))
    return None if hit is None else hit[2]


@_timed("pack_columns", "other")
def pack_columns(cols, out, out_col0=0):
    """cols: list of [B] or [B, w] device tensors -> out[:, out_col0:...] (int32 or float32)."""
    lib = _lib.load()
    B = out.shape[0]
    i = 0
    col = out_col0
    while i < len(cols):
        chunk = cols[i:i + _lib.FX_PACK_MAX_COLS]
        keep = []
        dts, ws = [], []
        for t in chunk:
            _need_cuda(t, "input column")
            if t.dtype not in _DT:
                raise _lib.FxError("unsupported input dtype %s" % t.dtype)
            if not t.is_contiguous():
                t = t.contiguous()
            if t.shape[0] != B:
                raise _lib.FxError("input column has %d rows, expected %d" % (t.shape[0], B))
            keep.append(t)
            dts.append(_DT[t.dtype])
            ws.append(1 if t.dim() == 1 else int(t.shape[1]))
        check(lib.fx_pack_columns(_lib.ptr_array(keep), _lib.i32_array(dts), _lib.i32_array(ws),
                                  len(keep), B, _DT[out.dtype], ptr(out), out.stride(0), col,
                                  stream_ptr(out.device)), "fx_pack_columns")
        col += sum(ws)
        i += len(chunk)
    return out


def _gather_bytes(table, D, ids, col_row_base, col_vocab, col_out_off, dense, num_w,
                  num_out_off, out, scal, n_cols=None):
    # algorithmic bytes: rows + ids + dense in, the [B,F,D] record out
    B = out.shape[0]
    C_ = 0 if ids is None else (ids.shape[1] if n_cols is None else n_cols)
    Fd = 0 if dense is None else dense.shape[1]
    return B * (C_ * (4 * D + 4) + Fd * 4 + (C_ + Fd) * 4 * D)


def _row_ld(table):
    """Row stride in elements of a [rows, D] table that may be a field of a row record (0: packed)."""
    if table is None or table.dim() != 2:
        return 0
    assert table.stride(1) == 1, "table rows must be contiguous"
    return int(table.stride(0))


@_timed("k_emb_gather_fwd", "sparse_path", _gather_bytes)
def emb_gather_fwd(table, D, ids, col_row_base, col_vocab, col_out_off, dense, num_w,
                   num_out_off, out, scal, n_cols=None):
    """n_cols: gather only the first n_cols id columns (the rest are pooled sequences)."""
    lib = _lib.load()
    B = out.shape[0]
    C_ = 0 if ids is None else (ids.shape[1] if n_cols is None else n_cols)
    Fd = 0 if dense is None else dense.shape[1]
    check(lib.fx_emb_gather_fwd(ptr(table), D, ptr(ids), 0 if ids is None else ids.stride(0),
                                ptr(col_row_base), ptr(col_vocab), ptr(col_out_off), C_,
                                ptr(dense), 0 if dense is None else dense.stride(0), ptr(num_w),
                                ptr(num_out_off), Fd, ptr(out), out.stride(0), B, ptr(scal),
                                _row_ld(table), stream_ptr(out.device)), "fx_emb_gather_fwd")
    return out


POOL_SUM, POOL_MEAN = 0, 1


@_timed("emb_seq_pool_fwd", "sparse_path")
def emb_seq_pool_fwd(table, D, ids, col_row_base, col_vocab, seq_col0, seq_len, seq_mode,
                     seq_out_off, out, denom, scal):
    """Pooled sequence features -> their slots of the record `out`; denom [B, n_seq] out."""
    lib = _lib.load()
    B, n_seq = denom.shape
    check(lib.fx_emb_seq_pool_fwd(ptr(table), D, ptr(ids), ids.stride(0), ptr(col_row_base),
                                  ptr(col_vocab), ptr(seq_col0), ptr(seq_len), ptr(seq_mode),
                                  ptr(seq_out_off), n_seq, ptr(out), out.stride(0), ptr(denom),
                                  B, ptr(scal), _row_ld(table), stream_ptr(out.device)),
          "fx_emb_seq_pool_fwd")
    return out


def dedup_workspace_bytes(n):
    lib = _lib.load()
    nbytes = lib.fx_dedup_workspace_bytes(n)
    if nbytes == 0:
        check(1, "fx_dedup_workspace_bytes")
    return int(nbytes)


class DedupResult(object):
    """Unique rows of a batch's lookups (device resident; count stays on the device)."""
    __slots__ = ("sorted_key", "sorted_pos", "uniq_row", "seg_start", "n_unique", "n_max", "C",
                 "sorted_uid")

    def __init__(self, n, C_, device, want_uid=False):
        self.sorted_uid = torch.empty(n, dtype=torch.int32, device=device) if want_uid else None
        self.sorted_key = torch.empty(n, dtype=torch.int32, device=device)
        self.sorted_pos = torch.empty(n, dtype=torch.int32, device=device)
        self.uniq_row = torch.empty(n, dtype=torch.int32, device=device)
        self.seg_start = torch.empty(n + 1, dtype=torch.int32, device=device)
        self.n_unique = torch.empty(1, dtype=torch.int32, device=device)   # always written by fx_dedup
        self.n_max = n
        self.C = C_


@_timed("dedup", "sparse_path")
def dedup(ids, col_row_base, col_vocab, col_pad, total_rows, workspace, result=None,
          n_shards=1, want_uid=False, columns_sorted=False, begin_scal=None, grouped=False):
    """grouped: ascending unique rows are not needed (fx_dedup's columns_sorted = 2: the bucketed in-LDS path)."""
    lib = _lib.load()
    B, C_ = ids.shape
    if result is None:
        result = DedupResult(B * C_, C_, ids.device, want_uid=want_uid)
    mode = 0
    if n_shards == 1:
        if columns_sorted and B <= 8192 and C_ <= 256:
            mode = 1
        elif grouped:
            mode = 2
    check(lib.fx_dedup(ptr(ids), ids.stride(0), B, C_, ptr(col_row_base), ptr(col_vocab),
                       ptr(col_pad), total_rows, ptr(workspace), workspace.numel(),
                       ptr(result.sorted_key), ptr(result.sorted_pos), ptr(result.uniq_row),
                       ptr(result.seg_start), ptr(result.n_unique), ptr(result.sorted_uid),
                       n_shards, mode, ptr(begin_scal),
                       stream_ptr(ids.device)),
          "fx_dedup")
    return result


@_timed("dedup_sorted_runs", "sparse_path")
def dedup_sorted_runs(ids, n_runs, vocab, pad, workspace, result=None):
    """De-dup of `n_runs` consecutive ascending runs of ids of one table (rows [0, vocab), `pad`
    marks empty entries at each run's tail): a merge by rank counting, no sort."""
    lib = _lib.load()
    n = ids.numel()
    assert ids.is_contiguous() and n % n_runs == 0
    if result is None:
        result = DedupResult(n, 1, ids.device)
    check(lib.fx_dedup_sorted_runs(ptr(ids), n_runs, n // n_runs, vocab, pad, ptr(workspace),
                                   workspace.numel(), ptr(result.sorted_key),
                                   ptr(result.sorted_pos), ptr(result.uniq_row),
                                   ptr(result.seg_start), ptr(result.n_unique),
                                   ptr(result.sorted_uid), stream_ptr(ids.device)),
          "fx_dedup_sorted_runs")
    return result


def shard_plan_workspace_ints(n_lookups, n_shards):
    return int(_lib.load().fx_shard_plan_workspace_ints(n_lookups, n_shards))


@_timed("shard_plan", "other")
def shard_plan(dd, n_shards, total_rows, cap, send_idx, uniq_slot, lookup_slot, scal,
               global_keys=False, workspace=None, slot_uniq=None):
    """slot_uniq [n_shards * cap] (out, optional): unique index held by every exchange slot, -1 = empty."""
    check(_lib.load().fx_shard_plan(ptr(dd.uniq_row), ptr(dd.n_unique), ptr(dd.sorted_pos),
                                    ptr(dd.sorted_uid), dd.n_max, n_shards, total_rows, cap,
                                    ptr(send_idx), ptr(uniq_slot), ptr(lookup_slot), ptr(scal),
                                    1 if global_keys else 0, ptr(workspace), ptr(slot_uniq),
                                    stream_ptr(send_idx.device)), "fx_shard_plan")


def _gx_arrays(tables):
    """tables: [(tensor or None, D, column offset)] -> host arrays of the exchange-block entry points."""
    n = len(tables)
    ptrs = _lib.ptr_array([t for t, _, _ in tables])
    return ptrs, (C.c_int32 * n)(*[int(d) for _, d, _ in tables]), \
        (C.c_int32 * n)(*[int(o) for _, _, o in tables]), n


@_timed("fill_grad_block", "other")
def fill_grad_block(tables, slot_uniq, block):
    """tables: [(G [n_max, D] or None, D, column offset)]; block [n_slots, ld] is written densely."""
    ptrs, ds, offs, n = _gx_arrays(tables)
    check(_lib.load().fx_fill_grad_block(ptrs, ds, offs, n, ptr(slot_uniq), block.shape[0], ptr(block),
                                         block.stride(0), stream_ptr(block.device)),
          "fx_fill_grad_block")


def owner_grad_reduce_partials(n_max):
    return int(_lib.load().fx_owner_grad_reduce_partials(n_max))


@_timed("owner_grad_reduce", "other")
def owner_grad_reduce(grecv, dd, tables, sq_partials):
    """tables: [(G_out [n_max, D] or None, D, column offset)] of the received block grecv [n, ld]."""
    ptrs, ds, offs, n = _gx_arrays(tables)
    check(_lib.load().fx_owner_grad_reduce(ptr(grecv), grecv.stride(0), ptr(dd.sorted_pos),
                                           ptr(dd.seg_start), ptr(dd.n_unique), dd.n_max, ptrs, ds,
                                           offs, n, ptr(sq_partials), stream_ptr(grecv.device)),
          "fx_owner_grad_reduce")


@_timed("scatter_rows", "other")
def scatter_rows(src, row_map, n_rows, n_max, D, dst):
    """dst: [*, >= D] (a column range of a wider block is fine: its row stride is used)."""
    check(_lib.load().fx_scatter_rows(ptr(src), ptr(row_map), ptr(n_rows), n_max, D, ptr(dst),
                                      dst.stride(0), stream_ptr(dst.device)), "fx_scatter_rows")


@_timed("split_rows", "other")
def split_rows(src, n_rows, parts, zero_tail_rows=0):
    """parts: [(column offset, dst [n_rows + zero_tail_rows, width] contiguous)], one launch."""
    n = len(parts)
    dsts = _lib.ptr_array([d for _, d in parts])
    offs = (C.c_int32 * n)(*[o for o, _ in parts])
    widths = (C.c_int32 * n)(*[d.shape[1] for _, d in parts])
    check(_lib.load().fx_split_rows(ptr(src), src.stride(0), n_rows, n, dsts, offs, widths,
                                    zero_tail_rows, stream_ptr(src.device)), "fx_split_rows")


@_timed("sum_parts", "other")
def sum_parts(parts, out):
    check(_lib.load().fx_sum_parts(_lib.ptr_array(parts), _lib.i64_array([p.numel() for p in parts]),
                                   len(parts), ptr(out), stream_ptr(out.device)), "fx_sum_parts")


def emb_grad_reduce_partials(n_max, D):
    return int(_lib.load().fx_emb_grad_reduce_partials(n_max, D))


def emb_grad_reduce_scratch_ints(n_max):
    return int(_lib.load().fx_emb_grad_reduce_scratch_ints(n_max))


@_timed("emb_grad_reduce", "sparse_path")
def emb_grad_reduce(dout, dout_ld, col_out_off, C_, D, dd, G, sq_partials, scratch,
                    col_denom=None, denom=None):
    """col_denom / denom: per-column index into the [B, n_seq] pooling denominators (mean-pooled
    sequence columns), see fx_emb_grad_reduce_scaled."""
    lib = _lib.load()
    if col_denom is None:
        check(lib.fx_emb_grad_reduce(ptr(dout), dout_ld, ptr(col_out_off), C_, D,
                                     ptr(dd.sorted_pos), ptr(dd.seg_start), ptr(dd.n_unique),
                                     dd.n_max, ptr(G), ptr(sq_partials), ptr(scratch),
                                     stream_ptr(dout.device)), "fx_emb_grad_reduce")
        return
    check(lib.fx_emb_grad_reduce_scaled(ptr(dout), dout_ld, ptr(col_out_off), ptr(col_denom),
                                        ptr(denom), denom.stride(0), C_, D, ptr(dd.sorted_pos),
                                        ptr(dd.seg_start), ptr(dd.n_unique), dd.n_max, ptr(G),
                                        ptr(sq_partials), ptr(scratch), stream_ptr(dout.device)),
          "fx_emb_grad_reduce_scaled")


@_timed("emb_numeric_grad", "other")
def emb_numeric_grad(dout, dout_ld, num_out_off, dense, D, dnum_w):
    lib = _lib.load()
    B, Fd = dense.shape
    ws = torch.empty(_lib.FX_NUMGRAD_CHUNKS * Fd * D, dtype=torch.float32, device=dout.device)
    check(lib.fx_emb_numeric_grad(ptr(dout), dout_ld, ptr(num_out_off), ptr(dense),
                                  dense.stride(0), Fd, D, B, ptr(dnum_w), ptr(ws),
                                  stream_ptr(dout.device)), "fx_emb_numeric_grad")


@_timed("opt_begin_step", "other")
def opt_begin_step(scal):
    check(_lib.load().fx_opt_begin_step(ptr(scal), stream_ptr(scal.device)), "fx_opt_begin_step")


@_timed("clip_coef", "other")
def clip_coef(parts, scal):
    """parts: list of fp32 device tensors holding partial sums of squared gradient norms."""
    lib = _lib.load()
    if len(parts) > _lib.FX_CLIP_MAX_PARTS:
        raise _lib.FxError("too many partial-norm arrays (%d)" % len(parts))
    check(lib.fx_clip_coef(_lib.ptr_array(parts), _lib.i64_array([p.numel() for p in parts]),
                           len(parts), ptr(scal), stream_ptr(scal.device)), "fx_clip_coef")


def _need_fp32_table(table, who):
    if table.dtype != torch.float32:
        raise _lib.FxError("%s is the fp32-only kernel of round 1 and got a %s table; bf16 tables go "
                           "through the RowState entry points (adam_catchup_rows / "
                           "sparse_update_multi)" % (who, table.dtype))


@_timed("sparse_adam", "sparse_path")
def sparse_adam(table, m, v, last_step, D, dd, G, scal):
    _need_fp32_table(table, "fx_sparse_adam")
    check(_lib.load().fx_sparse_adam(ptr(table), ptr(m), ptr(v), ptr(last_step), D,
                                     ptr(dd.uniq_row), ptr(dd.n_unique), dd.n_max, ptr(G),
                                     ptr(scal), stream_ptr(table.device)), "fx_sparse_adam")


@_timed("adam_catchup", "sparse_path")
def adam_catchup(table, m, v, last_step, D, dd, total_rows, upto_offset, scal):
    """dd=None: every row of the table (flush); else only the unique rows of dd."""
    _need_fp32_table(table, "fx_adam_catchup")
    lib = _lib.load()
    if dd is None:
        check(lib.fx_adam_catchup(ptr(table), ptr(m), ptr(v), ptr(last_step), D, vp(0), vp(0), 0,
                                  total_rows, upto_offset, ptr(scal), stream_ptr(table.device)),
              "fx_adam_catchup")
    else:
        check(lib.fx_adam_catchup(ptr(table), ptr(m), ptr(v), ptr(last_step), D,
                                  ptr(dd.uniq_row), ptr(dd.n_unique), dd.n_max, total_rows,
                                  upto_offset, ptr(scal), stream_ptr(table.device)),
              "fx_adam_catchup")


@_timed("sparse_sgd", "sparse_path")
def sparse_sgd(table, D, dd, G, scal, last_step=None):
    _need_fp32_table(table, "fx_sparse_sgd")
    check(_lib.load().fx_sparse_sgd(ptr(table), ptr(last_step), D, ptr(dd.uniq_row),
                                    ptr(dd.n_unique), dd.n_max, ptr(G), ptr(scal),
                                    stream_ptr(table.device)), "fx_sparse_sgd")


def reg_stats(x, scal, partials):
    """partials [3 * FX_REG_BLOCKS]: sum p^2 | sum |p| | sum r(p)^2 over the flat tensor x."""
    check(_lib.load().fx_reg_stats(ptr(x), x.numel(), ptr(scal), ptr(partials),
                                   stream_ptr(x.device)), "fx_reg_stats")


def reg_cross(table, D, dd, G, scal, partials):
    check(_lib.load().fx_reg_cross(ptr(table), D, ptr(dd.uniq_row), ptr(dd.n_unique), dd.n_max,
                                   ptr(G), ptr(scal), ptr(partials), stream_ptr(table.device)),
          "fx_reg_cross")


def reg_dense_update(table, m, v, last_step, D, adam, scal):
    check(_lib.load().fx_reg_dense_update(ptr(table), ptr(m), ptr(v), ptr(last_step),
                                          table.shape[0], D, 1 if adam else 0, ptr(scal),
                                          stream_ptr(table.device)), "fx_reg_dense_update")


def _chunks(n):
    i = 0
    while i < n:
        yield i, min(n, i + _lib.FX_MT_MAX)
        i += _lib.FX_MT_MAX


@_timed("mt_sqnorm", "other")
def mt_sqnorm(grads, sq_partials):
    """sq_partials: fp32 [len(grads) * FX_MT_BLOCKS]."""
    lib = _lib.load()
    sizes = [g.numel() for g in grads]
    for a, b in _chunks(len(grads)):
        out = sq_partials[a * _lib.FX_MT_BLOCKS:]
        check(lib.fx_mt_sqnorm(_lib.ptr_array(grads[a:b]), _lib.i64_array(sizes[a:b]), b - a,
                               ptr(out), stream_ptr(sq_partials.device)), "fx_mt_sqnorm")


@_timed("mt_adam", "other")
def mt_adam(params, grads, ms, vs, scal):
    lib = _lib.load()
    sizes = [p.numel() for p in params]
    for a, b in _chunks(len(params)):
        check(lib.fx_mt_adam(_lib.ptr_array(params[a:b]), _lib.ptr_array(grads[a:b]),
                             _lib.ptr_array(ms[a:b]), _lib.ptr_array(vs[a:b]),
                             _lib.i64_array(sizes[a:b]), b - a, ptr(scal),
                             stream_ptr(scal.device)), "fx_mt_adam")


@_timed("mt_sgd", "other")
def mt_sgd(params, grads, scal):
    lib = _lib.load()
    sizes = [p.numel() for p in params]
    for a, b in _chunks(len(params)):
        check(lib.fx_mt_sgd(_lib.ptr_array(params[a:b]), _lib.ptr_array(grads[a:b]),
                            _lib.i64_array(sizes[a:b]), b - a, ptr(scal),
                            stream_ptr(scal.device)), "fx_mt_sgd")


def fm_fwd(emb, F, D, addend, out):
    B = emb.shape[0]
    check(_lib.load().fx_fm_fwd(ptr(emb), emb.stride(0), F, D, ptr(addend), ptr(out), B,
                                stream_ptr(emb.device)), "fx_fm_fwd")
    return out


def fm_bwd(emb, F, D, g, demb, accumulate=False):
    B = emb.shape[0]
    check(_lib.load().fx_fm_bwd(ptr(emb), emb.stride(0), F, D, ptr(g), ptr(demb), demb.stride(0),
                                1 if accumulate else 0, B, stream_ptr(emb.device)), "fx_fm_bwd")
    return demb


@_timed("lr_fwd", "sparse_path")
def lr_fwd(table1, ids, col_row_base, col_vocab, dense, num_w1, bias, out, scal):
    B = out.shape[0]
    C_ = 0 if ids is None else ids.shape[1]
    Fd = 0 if dense is None else dense.shape[1]
    check(_lib.load().fx_lr_fwd(ptr(table1), ptr(ids), 0 if ids is None else ids.stride(0),
                                ptr(col_row_base), ptr(col_vocab), C_, ptr(dense),
                                0 if dense is None else dense.stride(0), ptr(num_w1), Fd,
                                ptr(bias), ptr(out), B, ptr(scal), _row_ld(table1),
                                stream_ptr(out.device)), "fx_lr_fwd")
    return out


def gemm_workspace_floats(M, N, split_k):
    """Floats of split-K workspace for an [M, N] product: split_k slabs of the product + split_k vectors of
    M partial row sums (the fused bias gradient), as include/fxctr.h's fx_gemm_problem comment says
    (ws[split_k * M * N + z * M + m]), + 64 floats of alignment slack."""
    return split_k * M * (N + 1) + 64


def gemm(A, B_, C_, transa=False, transb=False, bias=None, act=0, zout=None, mul=None, mask=None,
         add=None, split_k=1, workspace=None, rowsum=None):
    """C = epilogue(op(A) . op(B)).  A, B, C: 2-D fp32 with unit inner stride."""
    if KernelTimer.recording:
        M_, N_ = C_.shape
        K_ = A.shape[0] if transa else A.shape[1]
        KernelTimer.note("k_gemm_f32", "gemm %dx%dx%d" % (M_, N_, K_), 2.0 * M_ * N_ * K_, _gemm,
                         (A, B_, C_, transa, transb, bias, act, zout, mul, mask, add, split_k,
                          workspace, rowsum), {})
    return _gemm(A, B_, C_, transa, transb, bias, act, zout, mul, mask, add, split_k, workspace,
                 rowsum)


def _epilogue(bias=None, act=0, zout=None, mul=None, mask=None, add=None, rowsum=None):
    epi = _lib.GemmEpilogue()
    epi.bias = bias.data_ptr() if bias is not None else None
    epi.act = act
    if zout is not None:
        epi.zout, epi.ldz = zout.data_ptr(), zout.stride(0)
    if mul is not None:
        epi.mul, epi.ldmul = mul.data_ptr(), mul.stride(0)
    if mask is not None:
        epi.mask, epi.ldmask = mask.data_ptr(), mask.stride(0)
    if add is not None:
        epi.add, epi.ldadd = add.data_ptr(), add.stride(0)
    if rowsum is not None:
        epi.rowsum = rowsum.data_ptr()
    return epi


def gemm_dw_dx(dz, x, W, dW, dx, split_k=1, workspace=None, rowsum=None, mask=None, add=None):
    """The two gradient GEMMs of a Linear / CrossNet layer in ONE launch (fx_gemm_f32_batch):
    dW[N, K] = dz^T x (split-K, `rowsum` = column sums of dz = bias gradient) and
    dx[M, K] = dz W with the `mask` / `add` epilogue.  Same results as the two gemm() calls."""
    if KernelTimer.recording:
        M_, N_ = dz.shape
        K_ = x.shape[1]
        KernelTimer.note("k_gemm_f32", "gemm2 %dx%dx%d" % (M_, N_, K_), 4.0 * M_ * N_ * K_,
                         _gemm_dw_dx, (dz, x, W, dW, dx, split_k, workspace, rowsum, mask, add), {})
    return _gemm_dw_dx(dz, x, W, dW, dx, split_k, workspace, rowsum, mask, add)


class _GemmProblem(object):
    """One GEMM of a batch: keeps the tensors (and the epilogue struct the C struct points at) alive."""

    def __init__(self, A, B_, C_, transa, transb, split_k, workspace, epi, keep):
        M, N = C_.shape
        K = A.shape[0] if transa else A.shape[1]
        self.M, self.N, self.K = M, N, K
        self.epi, self.keep = epi, keep
        self.struct = _lib.GemmProblem(1 if transa else 0, 1 if transb else 0, M, N, K, ptr(A),
                                       A.stride(0), ptr(B_), B_.stride(0), ptr(C_), C_.stride(0),
                                       C.pointer(epi), split_k, ptr(workspace))
        self.device = C_.device


def gemm_problem(A, B_, C_, transa=False, transb=False, bias=None, act=0, zout=None, mul=None,
                 mask=None, add=None, split_k=1, workspace=None, rowsum=None):
    """-> one entry for gemm_batch(); same arguments as gemm().  split_k is the LARGEST number of K slabs
    `workspace` holds (gemm_workspace_floats(M, N, split_k)); the library chooses the actual split."""
    epi = _epilogue(bias, act, zout, mul, mask, add, rowsum)
    return _GemmProblem(A, B_, C_, transa, transb, split_k, workspace, epi,
                        (A, B_, C_, bias, zout, mul, mask, add, workspace, rowsum))


def gemm_batch(problems):
    """Independent GEMMs in as few launches as possible (fx_gemm_f32_batch): up to four aligned problems
    leave as ONE grid on 128-row tiles, two workgroups per CU."""
    if not problems:
        return
    if KernelTimer.recording:
        KernelTimer.note("k_gemm_f32", "gemmN " + "+".join("%dx%dx%d" % (p.M, p.N, p.K) for p in problems),
                         sum(2.0 * p.M * p.N * p.K for p in problems), _gemm_batch, (problems,), {})
    return _gemm_batch(problems)


def _gemm_batch(problems):
    arr = (_lib.GemmProblem * len(problems))()
    for i, p in enumerate(problems):
        arr[i] = p.struct
    check(_lib.load().fx_gemm_f32_batch(arr, len(problems), stream_ptr(problems[0].device)),
          "fx_gemm_f32_batch")


def _gemm_dw_dx(dz, x, W, dW, dx, split_k, workspace, rowsum, mask, add):
    M, N = dz.shape
    K = x.shape[1]
    e1, e2 = _epilogue(rowsum=rowsum), _epilogue(mask=mask, add=add)
    probs = (_lib.GemmProblem * 2)()
    probs[0] = _lib.GemmProblem(1, 0, N, K, M, ptr(dz), dz.stride(0), ptr(x), x.stride(0), ptr(dW),
                                dW.stride(0), C.pointer(e1), split_k, ptr(workspace))
    probs[1] = _lib.GemmProblem(0, 0, M, K, N, ptr(dz), dz.stride(0), ptr(W), W.stride(0), ptr(dx),
                                dx.stride(0), C.pointer(e2), 1, None)
    check(_lib.load().fx_gemm_f32_batch(probs, 2, stream_ptr(dz.device)), "fx_gemm_f32_batch")


def _gemm(A, B_, C_, transa, transb, bias, act, zout, mul, mask, add, split_k, workspace, rowsum):
    lib = _lib.load()
    M, N = C_.shape
    K = A.shape[0] if transa else A.shape[1]
    epi = _epilogue(bias, act, zout, mul, mask, add, rowsum)
    check(lib.fx_gemm_f32(1 if transa else 0, 1 if transb else 0, M, N, K, ptr(A), A.stride(0),
                          ptr(B_), B_.stride(0), ptr(C_), C_.stride(0), C.byref(epi), split_k,
                          ptr(workspace), stream_ptr(C_.device)), "fx_gemm_f32")
    return C_


@_timed("colsum", "other")
def colsum(X, out, workspace):
    M, N = X.shape
    check(_lib.load().fx_colsum(ptr(X), X.stride(0), M, N, ptr(out), ptr(workspace),
                                stream_ptr(X.device)), "fx_colsum")
    return out


@_timed("mask_mul", "other")
def mask_mul(dy, y, out):
    """dy: [rows, cols] with unit inner stride (a column slice of a wider tensor is read in place)."""
    rows, cols = dy.shape
    check(_lib.load().fx_mask_mul(ptr(dy), dy.stride(0), ptr(y), y.stride(0), ptr(out), rows, cols,
                                  stream_ptr(dy.device)), "fx_mask_mul")
    return out


@_timed("cross_bwd_prep", "other")
def cross_bwd_prep(dxn, x0, z, t, dx0, init, add_dxn):
    rows, cols = dxn.shape
    check(_lib.load().fx_cross_bwd_prep(ptr(dxn), dxn.stride(0), ptr(x0), ptr(z), ptr(t), ptr(dx0),
                                        rows, cols, 1 if init else 0, 1 if add_dxn else 0,
                                        stream_ptr(dxn.device)), "fx_cross_bwd_prep")


@_timed("sigmoid_bce", "other")
def sigmoid_bce(logit, y, prob=None, loss=None, dlogit=None):
    check(_lib.load().fx_sigmoid_bce(ptr(logit), ptr(y), logit.numel(), ptr(prob), ptr(loss),
                                     ptr(dlogit), stream_ptr(logit.device)), "fx_sigmoid_bce")


def head_train_workspace_floats(M, K):
    return int(_lib.load().fx_head_train_workspace(M, K))


def head_train_ok(h, W, out_add=None):
    """The shapes / alignments fx_head_train takes (else the caller keeps the three-kernel path)."""
    K = W.shape[1]
    return (W.shape[0] == 1 and K % 4 == 0 and 8 < K <= 2048 and h.dim() == 2 and h.stride(1) == 1
            and h.stride(0) % 4 == 0 and h.data_ptr() % 16 == 0 and W.is_contiguous()
            and W.data_ptr() % 16 == 0
            and (out_add is None or (out_add.numel() == h.shape[0] and out_add.dim() <= 2)))


@_timed("head_train", "other")
def head_train(h, W, bias, out_add, y, mask_from, root_scale, logit, dlogit, dz, dW, db, loss, workspace):
    """Head forward + sigmoid / BCE + head backward in one pass (fx_head_train).  h: [M, K]; W: [1, K];
    out_add / y / logit / dlogit: M elements; dz: [M, K] or None; dW: [1, K]; db: [1] or None; loss: [];
    mask_from: -1 no ReLU mask on dz, else the first masked column (0: all of them)."""
    M, K = h.shape
    check(_lib.load().fx_head_train(ptr(h), h.stride(0), ptr(W), ptr(bias), ptr(out_add),
                                    (out_add.stride(0) if out_add is not None else 0), ptr(y), M, K,
                                    int(mask_from), float(root_scale), ptr(logit), ptr(dlogit), ptr(dz),
                                    (dz.stride(0) if dz is not None else 0), ptr(dW), ptr(db), ptr(loss),
                                    ptr(workspace), stream_ptr(h.device)), "fx_head_train")


# ---- DIN attention / Dice ---------------------------------------------------------------------
def _k_strides(K):
    """K: [B, L, E] with unit stride in E (a strided view of the gather record is fine)."""
    if K.stride(2) != 1:
        K = K.contiguous()
    return K, K.stride(0), K.stride(1)


def din_concat_fwd(q, K, out):
    K, sb, sl = _k_strides(K)
    B, L, E = K.shape
    check(_lib.load().fx_din_concat_fwd(ptr(q), q.stride(0), ptr(K), sb, sl, B, L, E, ptr(out),
                                        stream_ptr(out.device)), "fx_din_concat_fwd")
    return out


def din_concat_bwd(dx, q, K, dq, dK):
    K, sb, sl = _k_strides(K)
    B, L, E = K.shape
    check(_lib.load().fx_din_concat_bwd(ptr(dx), ptr(q), q.stride(0), ptr(K), sb, sl, B, L, E,
                                        ptr(dq), ptr(dK), dK.stride(0), dK.stride(1), 0,
                                        stream_ptr(dx.device)), "fx_din_concat_bwd")


def din_pool_fwd(w, ids, K, out):
    K, sb, sl = _k_strides(K)
    B, L, E = K.shape
    check(_lib.load().fx_din_pool_fwd(ptr(w), ptr(ids), ids.stride(0), ptr(K), sb, sl, B, L, E,
                                      ptr(out), stream_ptr(out.device)), "fx_din_pool_fwd")
    return out


def din_pool_bwd(w, ids, K, dout, dw, dK):
    K, sb, sl = _k_strides(K)
    B, L, E = K.shape
    check(_lib.load().fx_din_pool_bwd(ptr(w), ptr(ids), ids.stride(0), ptr(K), sb, sl, ptr(dout),
                                      B, L, E, ptr(dw), ptr(dK), dK.stride(0), dK.stride(1),
                                      stream_ptr(dout.device)), "fx_din_pool_bwd")


def dice_workspace_floats(H):
    return int(_lib.load().fx_dice_workspace_floats(H))


def dice_fwd(Z, alpha, eps, momentum, training, running_mean, running_var, stats, Y, workspace):
    N, H = Z.shape
    check(_lib.load().fx_dice_fwd(ptr(Z), N, H, ptr(alpha), eps, momentum, 1 if training else 0,
                                  ptr(running_mean), ptr(running_var), ptr(stats), ptr(Y),
                                  ptr(workspace), stream_ptr(Z.device)), "fx_dice_fwd")
    return Y


def dice_bwd(Z, dY, alpha, eps, training, stats, dZ, dalpha, workspace):
    N, H = Z.shape
    check(_lib.load().fx_dice_bwd(ptr(Z), ptr(dY), N, H, ptr(alpha), eps, 1 if training else 0,
                                  ptr(stats), ptr(dZ), ptr(dalpha), ptr(workspace),
                                  stream_ptr(Z.device)), "fx_dice_bwd")


def dice_local_sums(Z, sums, workspace):
    N, H = Z.shape
    check(_lib.load().fx_dice_local_sums(ptr(Z), N, H, ptr(sums), ptr(workspace),
                                         stream_ptr(Z.device)), "fx_dice_local_sums")


def dice_fwd_from_sums(Z, alpha, eps, momentum, sums, n_total, running_mean, running_var, stats, Y):
    N, H = Z.shape
    check(_lib.load().fx_dice_fwd_from_sums(ptr(Z), N, H, ptr(alpha), eps, momentum, ptr(sums),
                                            n_total, ptr(running_mean), ptr(running_var),
                                            ptr(stats), ptr(Y), stream_ptr(Z.device)),
          "fx_dice_fwd_from_sums")
    return Y


def dice_bwd_local_sums(Z, dY, alpha, eps, stats, sums3, workspace):
    N, H = Z.shape
    check(_lib.load().fx_dice_bwd_local_sums(ptr(Z), ptr(dY), N, H, ptr(alpha), eps, ptr(stats),
                                             ptr(sums3), ptr(workspace), stream_ptr(Z.device)),
          "fx_dice_bwd_local_sums")


def dice_bwd_from_sums(Z, dY, alpha, eps, stats, sums3, n_total, dZ):
    N, H = Z.shape
    check(_lib.load().fx_dice_bwd_from_sums(ptr(Z), ptr(dY), N, H, ptr(alpha), eps, ptr(stats),
                                            ptr(sums3), n_total, ptr(dZ), stream_ptr(Z.device)),
          "fx_dice_bwd_from_sums")


# ---- DIN attention with the attention MLP fused in (fx_din_attn.hip) --------------------------------
DIN_ATTN_MAX_E, DIN_ATTN_MAX_H = 16, 64


def _din_attn_flops(n_eval):
    def work(q, K, W1, *a, **kw):
        B, L, E = K.shape
        return 2.0 * n_eval * B * L * 4 * E * W1.shape[0]
    return work


def din_attn_workspace_floats(B, L, E, H):
    return int(_lib.load().fx_din_attn_workspace_floats(B, L, E, H))


def _din_attn_head(q, K):
    K, sb, sl = _k_strides(K)
    B, L, E = K.shape
    return K, (ptr(q), q.stride(0), ptr(K), sb, sl, B, L, E)


@_timed("din_attn_stats", "din_attention", _din_attn_flops(0))
def din_attn_stats(q, K, W1, b1, sums, workspace, stats=None, momentum=0.0, running_mean=None,
                   running_var=None, num_batches_tracked=None):
    """sums[2H] = [sum h | sum h^2] over the B*L positions, h = W1 [q,k,q-k,q*k] + b1.  stats given (one
    rank): dice_stats_from_sums(training) over these B*L rows rides in the same launch."""
    K, head = _din_attn_head(q, K)
    H = W1.shape[0]
    check(_lib.load().fx_din_attn_stats(*head, ptr(W1), ptr(b1), H,
                                        ptr(sums), ptr(workspace), ptr(stats), momentum,
                                        ptr(running_mean) if stats is not None else None,
                                        ptr(running_var) if stats is not None else None,
                                        ptr(num_batches_tracked) if stats is not None else None,
                                        stream_ptr(q.device)),
          "fx_din_attn_stats")


def dice_stats_from_sums(sums, H, n_total, momentum, training, running_mean, running_var, stats,
                         num_batches_tracked=None):
    """num_batches_tracked (int64[1], training only): += 1 in the same launch."""
    check(_lib.load().fx_dice_stats_from_sums(ptr(sums) if sums is not None else None, H, n_total,
                                              momentum, 1 if training else 0, ptr(running_mean),
                                              ptr(running_var),
                                              ptr(num_batches_tracked) if training else None, ptr(stats),
                                              stream_ptr(stats.device)), "fx_dice_stats_from_sums")


@_timed("din_attn_fwd", "din_attention", _din_attn_flops(1))
def din_attn_fwd(q, K, W1, b1, alpha, eps, stats, W2, b2, mask, a_out, out):
    """a_out[B, L] = attention logits (before the mask), out[B, E] = sum_l a mask k."""
    K, head = _din_attn_head(q, K)
    H = W1.shape[0]
    check(_lib.load().fx_din_attn_fwd(*head, ptr(W1), ptr(b1), H, ptr(alpha), eps, ptr(stats),
                                      ptr(W2), ptr(b2), ptr(mask),
                                      mask.stride(0) if mask is not None else 0, ptr(a_out),
                                      ptr(out), out.stride(0), stream_ptr(q.device)),
          "fx_din_attn_fwd")
    return out


@_timed("din_attn_bwd_sums", "din_attention", _din_attn_flops(0))
def din_attn_bwd_sums(q, K, W1, b1, alpha, eps, stats, W2, mask, dout, da, sums5, workspace):
    """da[B, L] = mask (dout . k);  sums5[5H] = [dalpha | sum dzhat | sum dzhat*zhat | dW2 | db2, 0...]."""
    K, head = _din_attn_head(q, K)
    H = W1.shape[0]
    check(_lib.load().fx_din_attn_bwd_sums(*head, ptr(W1), ptr(b1), H, ptr(alpha), eps, ptr(stats),
                                           ptr(W2), ptr(mask),
                                           mask.stride(0) if mask is not None else 0, ptr(dout),
                                           dout.stride(0), ptr(da), ptr(sums5), ptr(workspace),
                                           stream_ptr(q.device)), "fx_din_attn_bwd_sums")


@_timed("din_attn_bwd", "din_attention", _din_attn_flops(2))
def din_attn_bwd(q, K, W1, b1, alpha, eps, training, stats, W2, mask, a_logit, dout, da, sums5,
                 n_total, dq, dK, dW1b1, workspace, dq_accumulate=False):
    """dq_accumulate: dq += (the rows already hold another share of the target's gradient)."""
    K, head = _din_attn_head(q, K)
    H = W1.shape[0]
    check(_lib.load().fx_din_attn_bwd(*head, ptr(W1), ptr(b1), H, ptr(alpha), eps,
                                      1 if training else 0, ptr(stats), ptr(W2), ptr(mask),
                                      mask.stride(0) if mask is not None else 0, ptr(a_logit),
                                      ptr(dout), dout.stride(0), ptr(da), ptr(sums5), n_total,
                                      ptr(dq), dq.stride(0), 1 if dq_accumulate else 0, ptr(dK),
                                      dK.stride(0), dK.stride(1),
                                      ptr(dW1b1), ptr(workspace), stream_ptr(q.device)),
          "fx_din_attn_bwd")


def dot_interact_fwd(emb, F, D, out, tail=0):
    """out[:, :P] = pairwise dots (rows of out may be wider: out.stride(0)); tail: see fx_dot_interact_fwd."""
    B = emb.shape[0]
    check(_lib.load().fx_dot_interact_fwd(ptr(emb), emb.stride(0), F, D, B, ptr(out), out.stride(0), tail,
                                          stream_ptr(emb.device)), "fx_dot_interact_fwd")
    return out


def dot_interact_bwd(emb, g, F, D, demb, tail=0):
    B = emb.shape[0]
    check(_lib.load().fx_dot_interact_bwd(ptr(emb), emb.stride(0), ptr(g), g.stride(0), tail, F, D, B,
                                          ptr(demb), demb.stride(0), stream_ptr(emb.device)),
          "fx_dot_interact_bwd")
    return demb


# ---- CIN ------------------------------------------------------------------------------------------
def cin_workgroups():
    return int(_lib.load().fx_cin_workgroups())


def cin_wimg_floats(F0, Mi, D, O):
    """Floats of the packed LDS images of W the matrix-core CIN kernels copy (0: this shape runs on
    the VALU kernels, no image)."""
    return int(_lib.load().fx_cin_wimg_floats(F0, Mi, D, O))


def cin_pack_w(layers, D):
    """layers: [(W [O, F0*Mi], F0, Mi, w_img)] (<= 4): every W laid out as its kernels' LDS images in ONE
    launch, once per step; cin_fwd / cin_bwd take the w_img."""
    import ctypes as C
    n = len(layers)
    Ws = (C.c_void_p * n)(*[ptr(t[0]) for t in layers])
    imgs = (C.c_void_p * n)(*[ptr(t[3]) for t in layers])
    F0s = (C.c_int32 * n)(*[t[1] for t in layers])
    Mis = (C.c_int32 * n)(*[t[2] for t in layers])
    Os = (C.c_int32 * n)(*[t[0].shape[0] for t in layers])
    check(_lib.load().fx_cin_pack_w(n, Ws, F0s, Mis, D, Os, imgs, stream_ptr(layers[0][0].device)),
          "fx_cin_pack_w")


def _cin_flops(X0, Xi, W, *a, **kw):
    # the compress product of one pass: 2 * O * F0 * Mi * D flops per sample
    return 2.0 * X0.shape[0] * W.shape[0] * X0.shape[1] * Xi.shape[1] * X0.shape[2]


@_timed("cin_fwd", "cin", _cin_flops)
def cin_fwd(X0, Xi, W, bias, Xn, pool, w_img=None):
    """X0 [B,F0,D], Xi [B,Mi,D] contiguous; W [O, F0*Mi]; Xn [B,O,D]; pool: [B,O] view (stride ok)."""
    B, F0, D = X0.shape
    Mi, O = Xi.shape[1], W.shape[0]
    check(_lib.load().fx_cin_fwd(ptr(X0), X0.stride(0), F0, ptr(Xi), Xi.stride(0), Mi, D, ptr(W),
                                 ptr(bias), O, ptr(Xn), ptr(pool),
                                 0 if pool is None else pool.stride(0), B, ptr(w_img),
                                 stream_ptr(X0.device)), "fx_cin_fwd")


@_timed("cin_bwd", "cin", lambda *a, **kw: 2.0 * _cin_flops(*a, **kw))     # dX and dW passes
def cin_bwd(X0, Xi, W, dXn, dpool, dX0, accumulate_dx0, dXi, partial, w_img=None):
    """partial: [G, O*F0*Mi + O] (a column slice of a wider [G, .] buffer is written in place)."""
    B, F0, D = X0.shape
    Mi, O = Xi.shape[1], W.shape[0]
    check(_lib.load().fx_cin_bwd(ptr(X0), X0.stride(0), F0, ptr(Xi), Xi.stride(0), Mi, D, ptr(W), O,
                                 ptr(dXn), ptr(dpool), 0 if dpool is None else dpool.stride(0),
                                 ptr(dX0), dX0.stride(0), 1 if accumulate_dx0 else 0, ptr(dXi),
                                 dXi.stride(0), ptr(partial), partial.stride(0), B, ptr(w_img),
                                 stream_ptr(X0.device)),
          "fx_cin_bwd")


# ---- evaluation metrics ---------------------------------------------------------------------------
def binary_metrics(y_pred, y_true):
    """(logloss, AUC) of float32 device vectors, as sklearn's log_loss / roc_auc_score on float64.
    One host sync (reads three 8-byte words)."""
    lib = _lib.load()
    y_pred = y_pred.reshape(-1).contiguous().float()
    y_true = y_true.reshape(-1).contiguous().float()
    n = y_pred.numel()
    nbytes = int(lib.fx_binary_metrics_workspace_bytes(n))
    if nbytes == 0:
        check(1, "fx_binary_metrics_workspace_bytes")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=y_pred.device)
    ll = torch.empty(1, dtype=torch.float64, device=y_pred.device)
    cnt = torch.empty(2, dtype=torch.int64, device=y_pred.device)
    check(lib.fx_binary_metrics(ptr(y_pred), ptr(y_true), n, ptr(ws), nbytes, ptr(ll), ptr(cnt),
                                stream_ptr(y_pred.device)), "fx_binary_metrics")
    s2, n_pos = (int(v) for v in cnt.tolist())
    n_neg = n - n_pos
    if n_pos == 0 or n_neg == 0:
        raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that "
                         "case.")
    auc = (s2 - n_pos * (n_pos + 1)) / (2 * n_pos * n_neg)     # exact integers, one rounding
    return float(ll.item()) / n, auc


# ---- fused sparse front end / back end (csrc/fx_fused.hip) ----------------------------------------
class RowState(object):
    """One packed table + its optimizer state (struct fx_row_state)."""
    __slots__ = ("table", "m", "v", "last_step", "G", "D")

    def __init__(self, table, m, v, last_step, D, G=None):
        self.table, self.m, self.v, self.last_step, self.D, self.G = table, m, v, last_step, D, G


def _row_states(states):
    arr = (_lib.RowState * max(len(states), 1))()
    for i, s in enumerate(states):
        arr[i].table = s.table.data_ptr()
        arr[i].m = s.m.data_ptr() if s.m is not None else None
        arr[i].v = s.v.data_ptr() if s.v is not None else None
        arr[i].last_step = s.last_step.data_ptr() if s.last_step is not None else None
        arr[i].G = s.G.data_ptr() if s.G is not None else None
        arr[i].D = int(s.D)
        arr[i].table_dtype = _lib.FX_BF16 if s.table.dtype == torch.bfloat16 else _lib.FX_F32
        # row strides: 0 (packed) unless the arrays are fields of a row record
        arr[i].table_ld = _row_ld(s.table)
        arr[i].m_ld = _row_ld(s.m)
        arr[i].v_ld = _row_ld(s.v)
        arr[i].last_ld = 0 if s.last_step is None else int(s.last_step.stride(0))
    return arr


@_timed("dedup_catchup", "sparse_path")
def dedup_catchup(ids, col_row_base, col_vocab, col_pad, workspace, states, scal, begin_scal=None,
                  upto_offset=-1, want_uid=False, result=None):
    """Column fast path of the de-dup + (optionally) the fused begin-step + the exact-mode catch-up
    of every table group in `states` (list of RowState); 2 launches.  -> DedupResult."""
    lib = _lib.load()
    B, C_ = ids.shape
    if result is None:
        result = DedupResult(B * C_, C_, ids.device, want_uid=want_uid)
    check(lib.fx_dedup_catchup(ptr(ids), ids.stride(0), B, C_, ptr(col_row_base), ptr(col_vocab),
                               ptr(col_pad), ptr(workspace), workspace.numel(),
                               ptr(result.sorted_key), ptr(result.sorted_pos), ptr(result.uniq_row),
                               ptr(result.seg_start), ptr(result.n_unique), ptr(result.sorted_uid),
                               ptr(begin_scal), _row_states(states), len(states), upto_offset,
                               ptr(scal), stream_ptr(ids.device)), "fx_dedup_catchup")
    return result


def _emb_fm_bytes(table, D, ids, col_row_base, col_vocab, col_out_off, dense, num_w, num_out_off,
                  out, scal, table1=None, **kw):
    B = out.shape[0]
    C_ = 0 if ids is None else ids.shape[1]
    Fd = 0 if dense is None else dense.shape[1]
    # rows + ids + dense in (+ the 4-byte first-order rows), the record out
    eb = 2 if (table is not None and table.dtype == torch.bfloat16) else 4
    return B * (C_ * (eb * D + 4) + Fd * 4 + (C_ * 4 if table1 is not None else 0)
                + (C_ + Fd) * 4 * D)


@_timed("k_emb_fm_fwd", "sparse_path", _emb_fm_bytes, alone=True)
def emb_fm_fwd(table, D, ids, col_row_base, col_vocab, col_out_off, dense, num_w, num_out_off, out,
               scal, table1=None, num_w1=None, bias1=None, lr_out=None, fm_out=None, fm_lr_out=None,
               S=None, zero_ranges=()):
    """Gather + numeric expansion (+ first-order term) (+ FM second-order term), one launch.
    zero_ranges: up to two (offset, floats) ranges of every record row that the launch clears (reserved
    slots a later kernel of the step fills).
    table / table1 may be column ranges of a wider block (their row stride is passed on): the rows a
    row-sharded exchange delivered are read in place."""
    lib = _lib.load()
    B = out.shape[0]
    C_ = 0 if ids is None else ids.shape[1]
    Fd = 0 if dense is None else dense.shape[1]
    tdt = _lib.FX_BF16 if (table is not None and table.dtype == torch.bfloat16) else _lib.FX_F32
    check(lib.fx_emb_fm_fwd(ptr(table), tdt, D, ptr(ids), 0 if ids is None else ids.stride(0),
                            ptr(col_row_base), ptr(col_vocab), ptr(col_out_off), C_, ptr(dense),
                            0 if dense is None else dense.stride(0), ptr(num_w), ptr(num_out_off),
                            Fd, ptr(out), out.stride(0), B, ptr(table1), ptr(num_w1), ptr(bias1),
                            ptr(lr_out), ptr(fm_out), ptr(fm_lr_out), ptr(S), ptr(scal),
                            0 if table is None else table.stride(0),
                            0 if table1 is None else table1.stride(0),
                            *[int(x) for r in (tuple(zero_ranges) + ((0, 0), (0, 0)))[:2] for x in r],
                            stream_ptr(out.device)), "fx_emb_fm_fwd")
    return out


def emb_fm_bwd_partials(n_lookups, D):
    return int(_lib.load().fx_emb_fm_bwd_partials(n_lookups, D))


def emb_fm_bwd_workspace_floats(n_lookups, D, Fd):
    return int(_lib.load().fx_emb_fm_bwd_workspace_floats(n_lookups, D, Fd))


@_timed("emb_fm_bwd", "sparse_path")
def emb_fm_bwd(drec, rec, S, g_fm, g_lr, col_out_off, C_, D, dd, G, sq_partials, G1, sq1_partials,
               dense, num_out_off, B, dnum_w, dnum_w1, dbias1, workspace):
    """Backward of emb_fm_fwd: unique-row gradients of the D-float table (and of the D=1 table),
    their squared-norm partials, numeric weight / LR bias gradients; 2 launches.  dd must carry
    sorted_uid (dedup_catchup(..., want_uid=True))."""
    Fd = 0 if dense is None else dense.shape[1]
    have = dd is not None and C_ > 0
    check(_lib.load().fx_emb_fm_bwd(
        ptr(drec), 0 if drec is None else drec.stride(0), ptr(rec),
        0 if rec is None else rec.stride(0), ptr(S), ptr(g_fm), ptr(g_lr), ptr(col_out_off), C_, D,
        ptr(dd.sorted_pos) if have else vp(0), ptr(dd.sorted_uid) if have else vp(0),
        ptr(dd.seg_start) if have else vp(0), ptr(dd.n_unique) if have else vp(0),
        dd.n_max if have else 0, ptr(G), ptr(sq_partials), ptr(G1), ptr(sq1_partials), ptr(dense),
        0 if dense is None else dense.stride(0), ptr(num_out_off), Fd, B, ptr(dnum_w), ptr(dnum_w1),
        ptr(dbias1), ptr(workspace), stream_ptr(workspace.device)), "fx_emb_fm_bwd")


def adam_catchup_all(state, total_rows, upto_offset, scal):
    """Exact-mode flush of one table (fp32 or bf16): every row up to step + upto_offset."""
    check(_lib.load().fx_adam_catchup_all(_row_states([state]), total_rows, upto_offset, ptr(scal),
                                          stream_ptr(scal.device)), "fx_adam_catchup_all")


@_timed("adam_catchup_rows", "sparse_path")
def adam_catchup_rows(states, dd, upto_offset, scal):
    """Exact-mode catch-up of the unique rows of `dd` in every table group of `states` (RowState; fp32
    or bf16 tables), FX_MAX_TABLES groups per launch."""
    lib = _lib.load()
    for i in range(0, len(states), _lib.FX_MAX_TABLES):
        part = states[i:i + _lib.FX_MAX_TABLES]
        check(lib.fx_adam_catchup_rows(_row_states(part), len(part), ptr(dd.uniq_row),
                                       ptr(dd.n_unique), dd.n_max, upto_offset, ptr(scal),
                                       stream_ptr(scal.device)), "fx_adam_catchup_rows")


@_timed("owner_fetch_rows", "sparse_path")
def owner_fetch_rows(states, offs, dd, send, catchup, scal, upto_offset=-1, zero_row=None):
    """Owner side of the row-sharded forward: (exact-mode catch-up +) gather of every table group in
    `states` (RowState list) into its columns `offs` of the send block [n, ld]; one launch.
    zero_row: a 1-D fp32 view cleared on the way (the pad row of the received-rows block)."""
    arr = (C.c_int32 * len(offs))(*[int(o) for o in offs])
    check(_lib.load().fx_owner_fetch_rows(_row_states(states), arr, len(states), ptr(dd.uniq_row),
                                          ptr(dd.seg_start), ptr(dd.sorted_pos), ptr(dd.n_unique),
                                          send.shape[0], ptr(send), send.stride(0),
                                          1 if catchup else 0, upto_offset, ptr(scal), ptr(zero_row),
                                          0 if zero_row is None else zero_row.numel(),
                                          stream_ptr(send.device)), "fx_owner_fetch_rows")


@_timed("sparse_update_multi", "sparse_path")
def sparse_update_multi(kind, states, dd, scal):
    """Row update (kind 'adam' | 'sgd') of every table group in `states` (RowState with .G)."""
    lib = _lib.load()
    fn = lib.fx_sparse_adam_multi if kind == "adam" else lib.fx_sparse_sgd_multi
    check(fn(_row_states(states), len(states), ptr(dd.uniq_row), ptr(dd.n_unique), dd.n_max,
             ptr(scal), stream_ptr(scal.device)), "fx_sparse_%s_multi" % kind)


def pack_columns_multi_prepare(items):
    """Validate `items` (see pack_columns_multi) and build the ctypes argument blocks ONCE.
    -> a callable that launches the cast on torch's current stream; it keeps the source and
    destination tensors alive.  A replayed training step on a batch it has already seen costs
    the launches alone (BaseModel's captured step: rank_model._GraphStep.fill)."""
    lib = _lib.load()
    if not items:
        return lambda: None
    B = items[0][1].shape[0]
    device = items[0][1].device
    calls, keep = [], []
    i = 0
    while i < len(items):
        chunk = items[i:i + _lib.FX_PACKM_MAX_COLS]
        cols, dts, ws, outs, odts, olds = [], [], [], [], [], []
        for t, out, col0 in chunk:
            _need_cuda(t, "input column")
            if t.dtype not in _DT or out.dtype not in (torch.int32, torch.float32):
                raise _lib.FxError("unsupported dtype %s -> %s" % (t.dtype, out.dtype))
            if not t.is_contiguous():
                t = t.contiguous()
            if t.shape[0] != B or out.shape[0] != B:
                raise _lib.FxError("input column has %d rows, expected %d" % (t.shape[0], B))
            cols.append(t)
            dts.append(_DT[t.dtype])
            ws.append(1 if t.dim() == 1 else int(t.shape[1]))
            outs.append(out.data_ptr() + col0 * out.element_size())
            odts.append(_DT[out.dtype])
            olds.append(out.stride(0))
            keep.append((t, out))
        oarr = (vp * len(outs))()
        for k, a in enumerate(outs):
            oarr[k] = a
        calls.append((_lib.ptr_array(cols), _lib.i32_array(dts), _lib.i32_array(ws), oarr,
                      _lib.i32_array(odts), _lib.i64_array(olds), len(cols), B))
        i += len(chunk)
    fn = lib.fx_pack_columns_multi

    def launch(_keep=keep):
        stream = stream_ptr(device)
        for c in calls:
            check(fn(*c, stream), "fx_pack_columns_multi")
    return launch


@_timed("pack_columns_multi", "other")
def pack_columns_multi(items):
    """items: list of (column tensor [B] | [B,w], destination matrix, first destination column);
    every column is cast into its destination (int32 or float32) in ONE launch per 96 columns."""
    pack_columns_multi_prepare(items)()
