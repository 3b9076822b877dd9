"""Native (MI355X) drop-ins for the reference's hot-path layers.

Same class names, constructor signatures, forward signatures and state_dict keys as
`fuxictr.pytorch.layers` (reference paths cited per class), but every op on the path is a call
into libfxctr.so (include/fxctr.h):

  FeatureEmbeddingDict / FeatureEmbedding   feature_embedding.py:30-297
      one packed [sum(V_f), D] table per embedding dim, ONE gather launch for all fields that
      writes the final [B, F, D] record (dict entries are views of it; dict2tensor returns the
      record itself), sparse backward to unique rows, sparse-row Adam/SGD.
  LogisticRegression / FactorizationMachine / InnerProductInteraction (product_sum, inner_product)
      logistic_regression.py:24-59, factorization_machine.py:25-59, inner_product.py:23-70
  DIN_Attention / Dice                      target_attention.py:26-92, activations.py:24-51
  MLP_Block                                 mlp_block.py:24-96  (one autograd node, fp32 MFMA GEMMs)
  CrossNetV2                                cross_net.py:95-129 (one autograd node, fused epilogue)

torch is used for device memory, streams, nn.Module bookkeeping and the autograd tape only.
"""
import os
from collections import OrderedDict
from functools import partial  # noqa: F401  (used by eval'ed initializer strings)

import weakref

import torch
from torch import nn

from . import _lib, ops

_DEFAULT_DEVICE = None
_DIST = None    # DistContext when tables are row-sharded over the ranks of one node


def set_dist_context(ctx):
    """Row-shard every table built from now on over ctx's ranks (set by BaseModel(shard='row'))."""
    global _DIST
    _DIST = ctx


_EMB_DTYPE = torch.float32


def set_emb_dtype(name):
    """Storage type of the D > 1 embedding tables built from now on (BaseModel kwarg `emb_dtype`):
    'fp32' (the reference's) or 'bf16' (opt-in: rows are read as bf16, every sum / the Adam state /
    the update arithmetic stay fp32, updated rows are rounded to nearest-even)."""
    global _EMB_DTYPE
    key = str(name or "fp32").lower()
    if key in ("fp32", "float32", "f32"):
        _EMB_DTYPE = torch.float32
    elif key in ("bf16", "bfloat16"):
        _EMB_DTYPE = torch.bfloat16
    else:
        raise ValueError("emb_dtype={} is not supported.".format(name))


def set_default_device(device):
    """Device on which native layers allocate their tables (set by BaseModel.__init__)."""
    global _DEFAULT_DEVICE
    _DEFAULT_DEVICE = torch.device(device) if device is not None else None


def _alloc_device():
    if _DEFAULT_DEVICE is not None:
        return _DEFAULT_DEVICE
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() \
        else torch.device("cpu")


def not_in_whitelist(element, whitelist=[]):
    """fuxictr/utils.py:209"""
    if not whitelist:
        return False
    if not isinstance(whitelist, list):
        return element != whitelist
    return element not in whitelist


def get_initializer(initializer):
    """fuxictr/pytorch/torch_utils.py:175-194 (string -> callable via eval)."""
    if isinstance(initializer, str):
        try:
            initializer = eval(initializer)
        except Exception:
            raise ValueError("initializer={} is not supported.".format(initializer))
    return initializer


class FeatureDict(dict):
    """Batch dict with a per-batch cache (packed id matrices, de-dup results) shared by the
    embedding layers of one model, so the main and the LR tables pack / sort the ids once."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.cache = {}


# ------------------------------------------------------------------------------------------------
# pooling encoders (torch ops; SURVEY §8f-3 "next" row) — fuxictr/pytorch/layers/pooling.py:23-73
# ------------------------------------------------------------------------------------------------
class MaskedAveragePooling(nn.Module):
    def forward(self, embedding_matrix, mask=None):
        sum_out = torch.sum(embedding_matrix, dim=1)
        if mask is None:
            mask = embedding_matrix.sum(dim=-1) != 0
        return sum_out / (mask.float().sum(-1, keepdim=True) + 1e-12)


class MaskedSumPooling(nn.Module):
    def forward(self, embedding_matrix):
        return torch.sum(embedding_matrix, dim=1)


# ------------------------------------------------------------------------------------------------
# packed tables
# ------------------------------------------------------------------------------------------------
class _TableView(nn.Module):
    """state_dict-compatible stand-in for the per-feature nn.Embedding: `.weight` is a
    [vocab_size, D] view of the packed table (key `embedding_layers.<feature>.weight`)."""

    def __init__(self, weight_view, padding_idx):
        super().__init__()
        self.weight = nn.Parameter(weight_view, requires_grad=True)
        self.padding_idx = padding_idx
        self.num_embeddings, self.embedding_dim = weight_view.shape

    def forward(self, ids):  # convenience only; the layer never goes through here
        raise RuntimeError("_TableView is storage only; call FeatureEmbeddingDict.forward")


class _NumericView(nn.Module):
    """Stand-in for nn.Linear(1, D, bias=False): `.weight` is a [D, 1] view of the packed
    numeric-weight matrix (key `embedding_layers.<feature>.weight`)."""

    def __init__(self, weight_view):
        super().__init__()
        self.weight = nn.Parameter(weight_view, requires_grad=True)
        self.bias = None
        self.in_features, self.out_features = 1, weight_view.shape[0]


class _Plan(object):
    """Launch plan of one table group for one set of present features."""
    pass


class _ShardExchange(object):
    """Per-batch routing state of the row-sharded path (see _TableGroup.shard_exchange_ids)."""
    grads = None         # backward: id(group) -> per-unique-key gradient waiting for the exchange


# table groups that route with the same id plan (a model's D=16 tables and its D=1 LR tables):
# their row / row-gradient exchanges travel in ONE all-to-all per direction
_SHARD_PEERS = {}


class _PendingGrad(object):
    __slots__ = ("dd", "G", "sq")

    def __init__(self, dd, G, sq):
        self.dd, self.G, self.sq = dd, G, sq


class _TableGroup(object):
    """All id/numeric features of one FeatureEmbeddingDict that share an embedding dim D."""

    def __init__(self, D, device):
        self.D = D
        self.device = device
        self.tables = OrderedDict()   # owning feature -> (row_base, vocab, padding_idx)
        self.alias = {}               # feature -> owning feature (share_embedding)
        self.widths = {}              # id feature -> 1 or max_len
        self.numeric = []             # numeric features, row order of num_w
        self.total_rows = 0
        self.table = None             # [total_rows, D]
        self.record = None            # [total_rows, W] fp32 row record [p | m | v | last_step | pad] once an
                                      # exact-mode Adam is attached (adopt_record): table / m / v / last_step
                                      # are then column ranges of it, row stride W
        self.owner = None             # weakref to the FeatureEmbeddingDict whose Parameters view the table
        self.num_w = None             # [len(numeric), D]
        self.plans = {}
        # optimizer attachment
        self.scal = None
        self.opt_kind = None          # None | "adam" | "sgd"
        self.exact = False
        self.m = self.v = self.last_step = None
        self.num_grad = None
        self.pending = []
        self._shard_consts = {}
        self._reduce_scratch = {}
        self._await_exchange = []     # row-gradient buffers waiting for their all-to-all
        self.dedup_ws = None
        self.opt = None               # the native optimizer this group is attached to
        # row sharding (owner = row % world, local row = row // world)
        self.dist = _DIST
        self.n_shards = _DIST.world if _DIST is not None else 1
        self.sharded = _DIST is not None      # (a 1-rank context is kept only by FX_SHARD_WORLD1)
        self.rank = _DIST.rank if _DIST is not None else 0
        self.rows_per_shard = 0
        self.a2a_factor = 1.5
        self.owner_ws = None

    # -- construction -----------------------------------------------------------------------
    def add_table(self, feature, vocab, padding_idx, width):
        self.tables[feature] = (self.total_rows, int(vocab), padding_idx)
        self.widths[feature] = width
        self.total_rows += int(vocab)

    def add_alias(self, feature, owner, width):
        self.alias[feature] = self.alias.get(owner, owner)
        self.widths[feature] = width

    def add_numeric(self, feature):
        self.numeric.append(feature)

    def allocate(self):
        if self.total_rows >= 2 ** 32 - 1:
            raise NotImplementedError("packed table with %d rows exceeds the 2^32-1 row limit "
                                      "of the sparse path" % self.total_rows)
        if self.total_rows > 0 and self.sharded:
            # local shard + one all-zero pad row (index rows_per_shard) that padded all-to-all
            # slots point at; it is never part of a de-dup result, so it is never updated.
            # (round 6: `emb_dtype: bf16` shards too — the owner widens its bf16 rows into the fp32 block of
            # the exchange, requesters read what they received exactly as an unsharded bf16 table's rows
            # are read after widening; moments, gradients and the update arithmetic are fp32 on the owner)
            self.rows_per_shard = -(-self.total_rows // self.n_shards)
            dt = _EMB_DTYPE if self.D > 1 else torch.float32
            self.table = torch.zeros(self.rows_per_shard + 1, self.D, dtype=dt, device=self.device)
        elif self.total_rows > 0:
            # (the D=1 tables of LogisticRegression stay fp32: 4-byte rows gain nothing from bf16)
            dt = _EMB_DTYPE if self.D > 1 else torch.float32
            self.table = torch.empty(self.total_rows, self.D, dtype=dt, device=self.device)
        if self.numeric:
            self.num_w = torch.empty(len(self.numeric), self.D, dtype=torch.float32,
                                     device=self.device)

    @staticmethod
    def record_width(D):
        """Floats per row record: p, m, v (D each) + the row's last_step, padded to 16 bytes, and to whole
        128-byte lines from 128 bytes up (D = 16: 64 floats = 256 B, two lines; D = 1: 4 floats = 16 B)."""
        w = -(-(3 * D + 1) // 4) * 4
        return w if w <= 32 else -(-w // 32) * 32

    def adopt_record(self):
        """Move the table into a [p | m | v | last_step] row record (exact-mode Adam, fp32 tables): a row's
        catch-up and its update touch ONE place in HBM instead of four arrays gigabytes apart
        (scripts/ubench/row_record.hip: 13.0 -> 5.9 us for 25 K rows of 33.76 M).  Every kernel that takes a
        table pointer takes its row stride.  Moments and stamps start at zero (a fresh optimizer)."""
        if self.table is None or self.table.dtype != torch.float32:
            return False
        if self.record is not None:
            self.record[:, self.D:].zero_()
            return True
        rec = torch.zeros(self.table.shape[0], self.record_width(self.D), dtype=torch.float32,
                          device=self.table.device)
        rec[:, :self.D].copy_(self.table)
        self._set_record(rec)
        return True

    def _set_record(self, rec):
        D = self.D
        self.record = rec
        self.table = rec[:, :D]
        self.m = rec[:, D:2 * D]
        self.v = rec[:, 2 * D:3 * D]
        self.last_step = rec.view(torch.int32)[:, 3 * D]
        owner = self.owner() if self.owner is not None else None
        if owner is not None:
            owner._bind_views()

    def drop_record(self):
        """Back to four packed arrays (a dtype change of the module: the record is fp32 by construction)."""
        if self.record is None:
            return
        t, m, v, l = (x.contiguous() for x in (self.table, self.m, self.v, self.last_step))
        self.record = None
        self.table, self.m, self.v, self.last_step = t, m, v, l
        owner = self.owner() if self.owner is not None else None
        if owner is not None:
            owner._bind_views()

    def table_of(self, feature):
        return self.tables[self.alias.get(feature, feature)]

    def local_range(self, feature):
        """Rows of this rank's shard that belong to `feature` (contiguous): [lo, hi)."""
        base, V, _ = self.table_of(feature)
        if not self.sharded:
            return base, base + V
        n, r = self.n_shards, self.rank
        lo = max(0, -(-(base - r) // n))
        hi = (base + V - 1 - r) // n + 1 if base + V - 1 >= r else 0
        return lo, max(lo, hi)

    def local_row(self, g):
        """Local index of global packed row g if this rank owns it, else None."""
        if not self.sharded:
            return g
        return g // self.n_shards if g % self.n_shards == self.rank else None

    def ensure_scal(self):
        if self.scal is None:
            self.scal = ops.new_scalars(self.device)
        return self.scal

    # -- plans ------------------------------------------------------------------------------
    def plan_for(self, ordered_features, pooled=None, tail=None, holes=(), tail_slots=0):
        """ordered_features: features of this group to embed, in feature_map order.
        pooled: {sequence feature: ops.POOL_SUM | ops.POOL_MEAN} — reduced inside the gather to ONE
        slot each (fx_emb_seq_pool_fwd); their id columns go last so that the plain gather takes
        the prefix [0, C_main).  tail: features whose id columns go last without being pooled here
        (the LR copy sums every column anyway and keeps the SAME column order as the embedding
        layer's plan, so both share one de-dup / one id exchange).  holes: raw sequence features
        whose positions are preceded by ONE reserved slot the gather leaves alone (DIN writes the
        attended vector there: [.. fields .., pooled, positions ..] makes the tower's input a prefix of
        the record and the record's gradient one buffer, no concatenation either way)."""
        pooled = pooled or {}
        last = set(pooled) | set(tail or ())
        holes = tuple(f for f in holes if f in self.widths and f not in pooled)
        key = (tuple(ordered_features), tuple(sorted(pooled.items())), tuple(sorted(last)), holes,
               int(tail_slots))
        plan = self.plans.get(key)
        if plan is not None:
            return plan
        p = _Plan()
        D = self.D
        p.num_feats, p.slot, p.pooled, p.hole = [], {}, dict(pooled), {}
        num_off = []
        slot = 0
        for f in ordered_features:           # slots follow the feature order
            if f in self.widths:
                w = 1 if f in pooled else self.widths[f]
                if f in holes:
                    p.hole[f] = slot
                    slot += 1
                p.slot[f] = (slot, w)
                slot += w
            else:
                p.num_feats.append(f)
                num_off.append(slot * D)
                p.slot[f] = (slot, 1)
                slot += 1
        ids_order = [f for f in ordered_features if f in self.widths and f not in last] + \
                    [f for f in ordered_features if f in self.widths and f in last]
        p.id_feats = [(f, self.widths[f]) for f in ids_order]
        row_base, vocab, pad, out_off, col_denom = [], [], [], [], []
        seq_col0, seq_len, seq_mode, seq_off = [], [], [], []
        p.C_main = 0
        for f, w in p.id_feats:
            base, V, pidx = self.table_of(f)
            s0 = p.slot[f][0]
            if f in pooled:
                if pooled[f] == ops.POOL_MEAN:
                    col_denom += [len(seq_col0)] * w
                else:
                    col_denom += [-1] * w
                seq_col0.append(len(row_base))
                seq_len.append(w)
                seq_mode.append(pooled[f])
                seq_off.append(s0 * D)
            else:
                col_denom += [-1] * w
            for k in range(w):
                row_base.append(base)
                vocab.append(V)
                pad.append(-1 if pidx is None else int(pidx))
                out_off.append((s0 if f in pooled else s0 + k) * D)
            if f not in pooled:
                assert p.C_main == len(row_base) - w, "pooled id columns must come last"
                p.C_main = len(row_base)
        # tail_slots: slots behind the last feature that the gather leaves alone (DLRM writes the bottom
        # tower's vector there: the interaction then reads ONE record, no concatenation)
        p.tail0 = slot if tail_slots else None
        slot += int(tail_slots)
        p.n_slots = slot
        # reserved slots (holes, tail) as (float offset, float count) ranges: the fused gather clears up to
        # two of them per record row (nobody may read uninitialised slots)
        rr = [(h * D, D) for h in sorted(p.hole.values())]
        if tail_slots:
            rr.append((p.tail0 * D, int(tail_slots) * D))
        p.reserved_ranges = tuple(rr[:2])
        p.C = len(row_base)
        p.n_seq = len(seq_col0)
        p.Fd = len(p.num_feats)
        dev = self.device
        p.col_row_base = torch.tensor(row_base, dtype=torch.int64, device=dev)
        p.col_vocab = torch.tensor(vocab, dtype=torch.int32, device=dev)
        p.col_pad = torch.tensor(pad, dtype=torch.int32, device=dev)
        p.col_out_off = torch.tensor(out_off, dtype=torch.int64, device=dev)
        p.col_zero_off = torch.zeros(max(p.C, 1), dtype=torch.int64, device=dev)
        p.num_out_off = torch.tensor(num_off, dtype=torch.int64, device=dev)
        p.num_zero_off = torch.zeros(max(p.Fd, 1), dtype=torch.int64, device=dev)
        p.seq_col0 = torch.tensor(seq_col0, dtype=torch.int32, device=dev)
        p.seq_len = torch.tensor(seq_len, dtype=torch.int32, device=dev)
        p.seq_mode = torch.tensor(seq_mode, dtype=torch.int32, device=dev)
        p.seq_out_off = torch.tensor(seq_off, dtype=torch.int64, device=dev)
        p.col_denom = torch.tensor(col_denom, dtype=torch.int32, device=dev) \
            if any(j >= 0 for j in col_denom) else None
        p.num_rows = [self.numeric.index(f) for f in p.num_feats]
        p.num_full = p.num_rows == list(range(len(self.numeric)))
        p.columns_sorted = all(w == 1 for _, w in p.id_feats) and \
            all(row_base[i] < row_base[i + 1] for i in range(len(row_base) - 1))
        p.sig = (tuple(p.id_feats), tuple(row_base), tuple(pad))
        p.pack_sig = (tuple(p.id_feats), tuple(p.num_feats))
        self.plans[key] = p
        return p

    # -- per-batch helpers ------------------------------------------------------------------
    def pack_inputs(self, plan, inputs):
        """-> (ids int32 [B, C] or None, dense fp32 [B, Fd] or None); cached on a FeatureDict."""
        cache = getattr(inputs, "cache", None)
        ckey = ("pack", plan.pack_sig)
        if cache is not None and ckey in cache:
            return cache[ckey]
        first = inputs[plan.id_feats[0][0] if plan.id_feats else plan.num_feats[0]]
        B = first.shape[0]
        ids = dense = None
        if plan.C:
            ids = torch.empty(B, plan.C, dtype=torch.int32, device=self.device)
            ops.pack_columns([inputs[f] for f, _ in plan.id_feats], ids)
        if plan.Fd:
            dense = torch.empty(B, plan.Fd, dtype=torch.float32, device=self.device)
            ops.pack_columns([inputs[f] for f in plan.num_feats], dense)
        if cache is not None:
            cache[ckey] = (ids, dense)
        return ids, dense

    def pack_dense(self, inputs, names):
        """Numeric input columns that are NOT features of this group as one fp32 [B, len(names)] block,
        through the same cast launch / cache as the group's own inputs (a captured step fills the
        block together with the id matrix, outside the graph).  DLRM's bottom-tower input."""
        key = tuple(names)
        plan = self._dense_plans.get(key) if hasattr(self, "_dense_plans") else None
        if plan is None:
            if not hasattr(self, "_dense_plans"):
                self._dense_plans = {}
            plan = _Plan()
            plan.id_feats, plan.num_feats, plan.C, plan.Fd = [], list(names), 0, len(names)
            plan.pack_sig = ((), key)
            self._dense_plans[key] = plan
        return self.pack_inputs(plan, inputs)[1]

    def row_state(self, G=None):
        return ops.RowState(self.table, self.m, self.v, self.last_step, self.D, G)

    def fast_columns(self, plan, B):
        """The column path of the de-dup (fx_dedup_catchup) applies: every id column owns its own
        table, in column order, unsharded, batch small enough for one in-LDS sort per column."""
        return (not self.sharded and plan.n_seq == 0 and plan.columns_sorted and B <= 8192
                and plan.C <= 256 and _lib.row_lanes(self.D) <= 64)

    def fused_front(self, plan):
        """The fused gather(+LR+FM) / balanced backward kernels apply: any plan whose id columns each
        fill one slot of the record (categorical columns and the positions of RAW sequences alike;
        pooled sequences keep the pooling kernels).  Row-sharded groups run the same kernels over the
        rows received from their owners (slot matrix instead of ids)."""
        return plan.n_seq == 0 and _lib.row_lanes(self.D) <= 64

    def dedup(self, plan, ids, inputs):
        cache = getattr(inputs, "cache", None)
        ckey = ("dedup", plan.sig, self.total_rows)
        if cache is not None and ckey in cache:
            return cache[ckey]
        n = ids.shape[0] * ids.shape[1]
        if self.dedup_ws is None or self.dedup_ws[0] != n:
            self.dedup_ws = (n, torch.empty(ops.dedup_workspace_bytes(n), dtype=torch.uint8,
                                            device=self.device))
        # (the optimizer step's device-side opening rides in the de-dup's first launch)
        begin = self.opt.take_begin() if self.opt is not None else None
        # (unsharded: nothing downstream needs ascending rows — sequence schemas take the bucketed in-LDS path)
        dd = ops.dedup(ids, plan.col_row_base, plan.col_vocab, plan.col_pad, self.total_rows,
                       self.dedup_ws[1], columns_sorted=plan.columns_sorted, want_uid=True,
                       begin_scal=begin, grouped=not self.sharded)
        if cache is not None:
            cache[ckey] = dd
        return dd

    def select_num_w(self, plan):
        if plan.Fd == 0:
            return None
        if plan.num_full:
            return self.num_w
        idx = torch.tensor(plan.num_rows, dtype=torch.int64, device=self.device)
        return self.num_w.index_select(0, idx)

    def prepare_train(self, plan, ids, inputs, peers=()):
        """De-dup the batch's rows; in exact mode bring them up to date before they are read.
        peers: other table groups that are looked up with the SAME id plan in this step (the D=1
        tables of LogisticRegression): their rows are caught up in the same launch."""
        if plan.C == 0 or self.opt_kind is None:
            return None
        if self.opt is not None:
            self.opt.ensure_begun()     # forward; backward; step(); zero_grad() loops: see optim.py
        cache = getattr(inputs, "cache", None)
        ckey = ("dedup", plan.sig, self.total_rows)
        todo = [g for g in (self,) + tuple(peers)
                if g.exact and g.opt_kind == "adam"
                and not (cache is not None and ("caughtup", id(g), ckey) in cache)]
        dd = cache.get(ckey) if cache is not None else None
        if dd is None and self.fast_columns(plan, ids.shape[0]):
            n = ids.shape[0] * ids.shape[1]
            if self.dedup_ws is None or self.dedup_ws[0] != n:
                self.dedup_ws = (n, torch.empty(ops.dedup_workspace_bytes(n), dtype=torch.uint8,
                                                device=self.device))
            begin = self.opt.take_begin() if self.opt is not None else None
            dd = ops.dedup_catchup(ids, plan.col_row_base, plan.col_vocab, plan.col_pad,
                                   self.dedup_ws[1], [g.row_state() for g in todo], self.scal,
                                   begin_scal=begin, want_uid=True)
            if cache is not None:
                cache[ckey] = dd
                for g in todo:
                    cache[("caughtup", id(g), ckey)] = True
            return dd
        if dd is None:
            dd = self.dedup(plan, ids, inputs)
        if todo:
            # generic de-dup (sequence columns that alias a table, B > 8192): one dtype-aware launch
            # for every table group of the id plan (fx_adam_catchup is fp32-only)
            ops.adam_catchup_rows([g.row_state() for g in todo], dd, -1, self.scal)
            if cache is not None:
                for g in todo:
                    cache[("caughtup", id(g), ckey)] = True
        return dd

    def backward(self, plan, ids, dense, dout, dout_ld, col_off, num_off, dd, inputs_cache,
                 sx=None, denom=None):
        """Sparse + numeric gradients of one forward call (dout: grad of the output record)."""
        D = self.D
        if plan.Fd:
            g = torch.empty(plan.Fd, D, dtype=torch.float32, device=self.device)
            ops.emb_numeric_grad(dout, dout_ld, num_off, dense, D, g)
            self.add_num_grad(plan, g)
        col_denom = plan.col_denom if denom is not None else None
        if plan.C and sx is not None:
            self.shard_backward(plan, sx, dout, dout_ld, col_off, col_denom, denom)
        elif plan.C:
            if dd is None:
                dd = self.dedup(plan, ids, inputs_cache)
            G = torch.empty(dd.n_max, D, dtype=torch.float32, device=self.device)
            sq = torch.empty(ops.emb_grad_reduce_partials(dd.n_max, D), dtype=torch.float32,
                             device=self.device)
            ops.emb_grad_reduce(dout, dout_ld, col_off, plan.C, D, dd, G, sq,
                                self.reduce_scratch(dd.n_max), col_denom, denom)
            self.pending.append(_PendingGrad(dd, G, sq))

    def add_num_grad(self, plan, g):
        """g [plan.Fd, D]: gradient of the numeric weights used by `plan`."""
        if plan.num_full and self.num_grad is None:
            self.num_grad = g
        else:
            if self.num_grad is None:
                self.num_grad = torch.zeros_like(self.num_w)
            idx = torch.tensor(plan.num_rows, dtype=torch.int64, device=self.device)
            self.num_grad.index_add_(0, idx, g)

    def reduce_scratch(self, n_max):
        """Persistent scratch of fx_emb_grad_reduce (word 0 zero on entry, left zero on return)."""
        buf = self._reduce_scratch.get(n_max)
        if buf is None:
            buf = self._reduce_scratch[n_max] = torch.zeros(
                ops.emb_grad_reduce_scratch_ints(n_max), dtype=torch.int32, device=self.device)
        return buf

    # -- row-sharded exchange --------------------------------------------------------------
    def a2a_cap(self, n_lookups):
        per_peer = -(-n_lookups // self.n_shards)
        return int(-(-int(per_peer * self.a2a_factor) // 64) * 64 + 64)

    def shard_exchange_ids(self, plan, ids, inputs, track=False):
        """De-dup the local lookups owner-major, route the unique keys to their owners (one
        all-to-all) and de-dup what this rank received as an owner.  Shared by table groups with
        the same id columns / row bases (the D=16 and the D=1 LR tables)."""
        cache = getattr(inputs, "cache", None)
        ckey = ("shard", plan.sig, self.total_rows, self.n_shards)
        peers = _SHARD_PEERS.setdefault((plan.sig, self.total_rows, self.n_shards, id(self.dist)), [])
        if not any(p is self for p in peers):
            peers.append(self)
        if cache is not None and ckey in cache:
            return cache[ckey]
        if self.opt is not None and track:
            self.opt.ensure_begun()
        dev, N = self.device, self.n_shards
        n = ids.shape[0] * ids.shape[1]
        if self.dedup_ws is None or self.dedup_ws[0] != n:
            self.dedup_ws = (n, torch.empty(ops.dedup_workspace_bytes(n), dtype=torch.uint8,
                                            device=dev))
        # unique GLOBAL rows in ascending order (the column fast path applies); owners and slots are
        # derived by counting in fx_shard_plan(global_keys), no owner-major device sort
        begin = self.opt.take_begin() if self.opt is not None else None
        dd = ops.dedup(ids, plan.col_row_base, plan.col_vocab, plan.col_pad, self.total_rows,
                       self.dedup_ws[1], want_uid=True, columns_sorted=plan.columns_sorted,
                       begin_scal=begin)
        cap = self.a2a_cap(n)
        sx = _ShardExchange()
        sx.dd, sx.cap = dd, cap
        sx.key = (plan.sig, self.total_rows, N, id(self.dist))
        sx.rows = {}                                  # id(group) -> fetched rows of this batch
        sx.send_idx = torch.empty(N * cap, dtype=torch.int32, device=dev)
        sx.uniq_slot = torch.empty(n, dtype=torch.int32, device=dev)
        sx.lookup_slot = torch.empty(ids.shape[0], ids.shape[1], dtype=torch.int32, device=dev)
        sx.slot_uniq = torch.empty(N * cap, dtype=torch.int32, device=dev)   # slot -> unique key (-1)
        wkey = ("plan_ws", n, N)
        pws = self._shard_consts.get(wkey)
        if pws is None:
            pws = self._shard_consts[wkey] = torch.empty(ops.shard_plan_workspace_ints(n, N),
                                                         dtype=torch.int32, device=dev)
        ops.shard_plan(dd, N, self.total_rows, cap, sx.send_idx, sx.uniq_slot, sx.lookup_slot,
                       self.ensure_scal(), global_keys=True, workspace=pws, slot_uniq=sx.slot_uniq)
        if A2A_FILL_PROBE["on"]:
            # diagnostic (bench.py --probe-loss, eager steps only: this reads the device): how full the fullest
            # per-owner bucket of the exchange is, against its fixed capacity (a2a_factor x the even share)
            used = int((sx.slot_uniq.view(N, cap) >= 0).sum(dim=1).max().item())
            A2A_FILL_PROBE["max_used"] = max(A2A_FILL_PROBE["max_used"], used)
            A2A_FILL_PROBE["cap"] = cap
            A2A_FILL_PROBE["even_share"] = -(-n // N)
        sx.recv_idx = self.dist.all_to_all(sx.send_idx).view(N * cap, 1)
        sx.grads = {}                                 # id(group) -> per-unique-key gradient (backward)
        if self.owner_ws is None or self.owner_ws[0] != N * cap:
            self.owner_ws = (N * cap, torch.empty(ops.dedup_workspace_bytes(N * cap),
                                                  dtype=torch.uint8, device=dev))
        rps = self.rows_per_shard
        C = ids.shape[1]
        consts = self._shard_consts.get((C, cap))
        if consts is None:      # built once, outside any graph capture (H2D copies)
            consts = self._shard_consts[(C, cap)] = (
                torch.zeros(1, dtype=torch.int64, device=dev),
                torch.tensor([rps + 1], dtype=torch.int32, device=dev),
                torch.tensor([rps], dtype=torch.int32, device=dev),
                # lookups gather from the received-rows buffer [N*cap + 1, D] (last row = zeros)
                torch.zeros(C, dtype=torch.int64, device=dev),
                torch.full((C,), N * cap + 1, dtype=torch.int32, device=dev))
        sx.own_base, sx.own_vocab, sx.own_pad, sx.slot_base, sx.slot_vocab = consts
        # what arrived is N ascending runs (each peer's unique rows of this shard, pad rows at the
        # tail): merged by rank counting, no device sort
        sx.owner_dd = ops.dedup_sorted_runs(sx.recv_idx, N, rps + 1, rps, self.owner_ws[1])
        if cache is not None:
            cache[ckey] = sx
        return sx

    def shard_fetch_rows(self, sx, track):
        """Owner side: bring the requested rows up to date (exact mode) and gather them — every table
        group routed by this id plan in ONE launch, straight into its columns of one send block
        ([N*cap, 16 + 1 (+3 pad)]) — then one all-to-all.  -> this group's columns of the received block
        [N*cap + 1, width] (a strided view: the gather kernels read it in place; the last row is the
        all-zero pad slot), in the slot order of sx.lookup_slot."""
        N, cap = self.n_shards, sx.cap
        got = sx.rows.pop(id(self), None)
        if got is not None:
            return got                                # fetched together with a peer group
        layout, width = self.shard_layout(sx)
        send = torch.empty(N * cap, width, dtype=torch.float32, device=self.device)
        recv = torch.empty(N * cap + 1, width, dtype=torch.float32, device=self.device)
        catchup = bool(track) and all(p.exact and p.opt_kind == "adam" for p, _ in layout)
        if track and not catchup:
            for p, _ in layout:                       # mixed optimizers: separate catch-up launches
                if p.exact and p.opt_kind == "adam":
                    ops.adam_catchup_rows([p.row_state()], sx.owner_dd, -1, p.scal)     # (dtype-aware)
        ops.owner_fetch_rows([p.row_state() for p, _ in layout], [off for _, off in layout],
                             sx.owner_dd, send, catchup, self.ensure_scal(), zero_row=recv[N * cap])
        self.dist.all_to_all(send, recv=recv[:N * cap])
        for p, off in layout:
            sx.rows[id(p)] = recv[:, off:off + p.D]
        return sx.rows.pop(id(self))

    def shard_layout(self, sx):
        """-> ([(group, column offset)], block width): the table groups that share this exchange,
        float4-wide groups first so their columns stay 16-byte aligned, row stride a multiple of 4
        floats (vector loads / stores on the rows of the block).  Identical on every rank."""
        peers = [p for p in _SHARD_PEERS.get(sx.key, []) if p.table is not None]
        if not any(p is self for p in peers):
            peers = [self]
        peers = sorted(peers, key=lambda p: 0 if p.D % 4 == 0 else 1)
        layout, off = [], 0
        for p in peers:
            layout.append((p, off))
            off += p.D
        return layout, (off if len(peers) == 1 else -(-off // 4) * 4)

    def shard_backward(self, plan, sx, dout, dout_ld, col_off, col_denom=None, denom=None):
        """Requester: reduce to local unique keys, ship to owners; owner: reduce across ranks."""
        N, cap, D = self.n_shards, sx.cap, self.D
        dd = sx.dd
        G_loc = torch.empty(dd.n_max, D, dtype=torch.float32, device=self.device)
        sq = torch.empty(ops.emb_grad_reduce_partials(dd.n_max, D), dtype=torch.float32,
                         device=self.device)
        ops.emb_grad_reduce(dout, dout_ld, col_off, plan.C, D, dd, G_loc, sq,
                            self.reduce_scratch(dd.n_max), col_denom, denom)
        self.shard_send_grads(sx, G_loc)

    def shard_send_grads(self, sx, G_loc):
        """Requester: per-unique-key gradients G_loc [n_max, D] of this group; the exchange itself runs
        after autograd returns (finish_shard_backward, called by the optimizer on the main thread):
        collectives stay in one fixed program order on every rank."""
        sx.grads[id(self)] = G_loc
        self._await_exchange.append(sx)

    def finish_backward(self):
        """Owner side of the sharded backward: ship row gradients to their owners, reduce the
        contributions of all ranks per owned row."""
        finish_shard_backward([self])

    def flush(self):
        """exact mode: replay pending zero-gradient Adam steps for EVERY row (before eval/save)."""
        if self.exact and self.opt_kind == "adam" and self.table is not None:
            rows = self.rows_per_shard + 1 if self.sharded else self.total_rows
            if self.table.dtype == torch.bfloat16 or self.record is not None:
                ops.adam_catchup_all(self.row_state(), rows, 0, self.scal)
            else:
                ops.adam_catchup(self.table, self.m, self.v, self.last_step, self.D, None,
                                 rows, 0, self.scal)


def finish_shard_backward(groups):
    """Row-gradient exchange of every table group that has one waiting.  Groups that were routed by the
    same _ShardExchange travel side by side in ONE block ([N*cap, sum D]): one launch writes it (every
    slot: its unique key's gradient rows or zeros), one all-to-all ships it, one launch reduces what
    arrived per owned row for all groups and emits the squared-norm partials of the clip."""
    by_sx = OrderedDict()
    for grp in groups:
        waiting, grp._await_exchange = grp._await_exchange, []
        for sx in waiting:
            by_sx.setdefault(id(sx), (sx, []))[1].append(grp)
    for sx, grps in by_sx.values():
        first = grps[0]
        layout, width = first.shard_layout(sx)
        n = first.n_shards * sx.cap
        dev = first.device
        grads, sx.grads = sx.grads, {}
        block = torch.empty(n, width, dtype=torch.float32, device=dev)
        ops.fill_grad_block([(grads.get(id(p)), p.D, off) for p, off in layout], sx.slot_uniq, block)
        recv = first.dist.all_to_all(block)
        odd = sx.owner_dd
        outs = []
        for p, off in layout:
            G_own = torch.empty(odd.n_max, p.D, dtype=torch.float32, device=dev) \
                if id(p) in grads else None
            outs.append((G_own, p.D, off))
        sq = torch.empty(ops.owner_grad_reduce_partials(odd.n_max), dtype=torch.float32, device=dev)
        ops.owner_grad_reduce(recv, odd, outs, sq)
        for (p, _), (G_own, _, _) in zip(layout, outs):
            if G_own is not None:
                # (the partials cover every group of the block: they ride with the first one)
                p.pending.append(_PendingGrad(odd, G_own, sq))
                sq = None


class _EmbGatherFn(torch.autograd.Function):
    """All id columns + numeric columns of a group -> [B, n_slots * D]: one gather launch, plus one
    pooling launch when sequence features are reduced on the fly (plan.pooled)."""

    @staticmethod
    def forward(ctx, anchor, group, plan, ids, dense, dd, inputs, track):
        B = (ids if ids is not None else dense).shape[0]
        D = group.D
        out = torch.empty(B, plan.n_slots * D, dtype=torch.float32, device=group.device)
        sx = None
        if group.sharded and plan.C:
            # row-sharded: ids -> owners, rows <- owners, then the same kernels read the received
            # rows through the per-lookup slot matrix
            sx = group.shard_exchange_ids(plan, ids, inputs, track)
            # (these kernels read packed [rows, D] tables: this group's columns of the received block
            # are copied out; the fused front end — _EmbFMFn — reads the block in place)
            table = group.shard_fetch_rows(sx, track).contiguous()
            src, base, vocab = sx.lookup_slot, sx.slot_base, sx.slot_vocab
        else:
            table, src, base, vocab = group.table, ids, plan.col_row_base, plan.col_vocab
        ops.emb_gather_fwd(table, D, src, base, vocab, plan.col_out_off, dense,
                           group.select_num_w(plan), plan.num_out_off, out, group.ensure_scal(),
                           n_cols=plan.C_main if plan.n_seq else None)
        denom = None
        if plan.n_seq:
            denom = torch.empty(B, plan.n_seq, dtype=torch.float32, device=group.device)
            ops.emb_seq_pool_fwd(table, D, src, base, vocab, plan.seq_col0, plan.seq_len,
                                 plan.seq_mode, plan.seq_out_off, out, denom, group.ensure_scal())
        ctx.group, ctx.plan, ctx.ids, ctx.dense, ctx.dd, ctx.sx = group, plan, ids, dense, dd, sx
        ctx.denom = denom if plan.col_denom is not None else None
        ctx.inputs = inputs if hasattr(inputs, "cache") else None
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous()
        ctx.group.backward(ctx.plan, ctx.ids, ctx.dense, dout, dout.stride(0),
                           ctx.plan.col_out_off, ctx.plan.num_out_off, ctx.dd, ctx.inputs,
                           sx=ctx.sx, denom=ctx.denom)
        return None, None, None, None, None, None, None, None


class _EmbFMFn(torch.autograd.Function):
    """The fused sparse front end (csrc/fx_fused.hip): gather + numeric expansion, and — when the
    model has them — the first-order term of LogisticRegression (logistic_regression.py:46-59) and
    the FM second-order term (inner_product.py:55-62), ONE launch; outputs = (record [B, F*D],
    lr_out, fm_out, fm_lr_out = fm + lr), the last three [B,1] or None.  Backward: ONE balanced
    run-reduce for the D-float rows and the D=1 rows with the FM gradient folded in and the clip-norm
    partials fused, plus one launch for the numeric weights / LR bias."""

    @staticmethod
    def forward(ctx, anchor, lr_anchor, lr_bias, group, plan, lr_group, lr_plan, ids, dense, dd,
                inputs, want_fm, track=True):
        ctx.set_materialize_grads(False)
        B = (ids if ids is not None else dense).shape[0]
        D = group.D
        dev = group.device
        out = torch.empty(B, plan.n_slots * D, dtype=torch.float32, device=dev)
        want_lr = lr_group is not None
        sx = None
        table, table1 = group.table, (lr_group.table if want_lr else None)
        g_ids, g_base, g_vocab = ids, plan.col_row_base, plan.col_vocab
        if group.sharded and plan.C:
            # row-sharded tables: ids -> owners, rows back (ONE all-to-all for the D-float and the
            # D=1 rows), then the same kernel reads the received rows through the slot matrix
            sx = group.shard_exchange_ids(plan, ids, inputs, track)
            if want_lr:
                lr_group.shard_exchange_ids(lr_plan, ids, inputs, track)     # joins the row exchange
            table = group.shard_fetch_rows(sx, track)
            table1 = lr_group.shard_fetch_rows(sx, track) if want_lr else None
            g_ids, g_base, g_vocab = sx.lookup_slot, sx.slot_base, sx.slot_vocab
            if os.environ.get("FX_DEBUG_FUSED_SHARD") == "1":
                print("[fx] fused front end on row-sharded tables (lr=%s fm=%s)" % (want_lr, want_fm),
                      flush=True)
        lr_out = torch.empty(B, 1, dtype=torch.float32, device=dev) if want_lr else None
        fm_out = torch.empty(B, 1, dtype=torch.float32, device=dev) if want_fm else None
        fm_lr = torch.empty(B, 1, dtype=torch.float32, device=dev) if (want_fm and want_lr) else None
        S = torch.empty(B, D, dtype=torch.float32, device=dev) if want_fm else None
        ops.emb_fm_fwd(table, D, g_ids, g_base, g_vocab, plan.col_out_off,
                       dense, group.select_num_w(plan), plan.num_out_off, out, group.ensure_scal(),
                       table1=table1,
                       num_w1=lr_group.select_num_w(lr_plan) if want_lr else None,
                       bias1=lr_bias if want_lr else None, lr_out=lr_out, fm_out=fm_out,
                       fm_lr_out=fm_lr, S=S, zero_ranges=plan.reserved_ranges)
        ctx.group, ctx.plan, ctx.lr_group, ctx.lr_plan = group, plan, lr_group, lr_plan
        ctx.ids, ctx.dense, ctx.dd, ctx.out, ctx.S = ids, dense, dd, out, S
        ctx.sx = sx
        ctx.inputs = inputs if hasattr(inputs, "cache") else None
        ctx.has_bias = lr_bias is not None
        return out, lr_out, fm_out, fm_lr

    @staticmethod
    def backward(ctx, d_out, d_lr, d_fm, d_fm_lr):
        group, plan, lr_group, lr_plan = ctx.group, ctx.plan, ctx.lr_group, ctx.lr_plan

        def both(a, b):
            if a is None:
                return b
            return a if b is None else a + b
        g_fm, g_lr = both(d_fm, d_fm_lr), both(d_lr, d_fm_lr)
        none = (None,) * 13
        if d_out is None and g_fm is None and g_lr is None:
            return none
        D, dev = group.D, group.device
        B = ctx.out.shape[0]
        d_out = d_out.contiguous() if d_out is not None else None
        g_fm = g_fm.contiguous() if g_fm is not None else None
        g_lr = g_lr.contiguous() if g_lr is not None else None
        dd = ctx.sx.dd if ctx.sx is not None else ctx.dd
        if plan.C and (dd is None or dd.sorted_uid is None):
            dd = group.dedup(plan, ctx.ids, ctx.inputs)      # (no optimizer attached)
        G = sq = G1 = sq1 = None
        if plan.C:
            G = torch.empty(dd.n_max, D, dtype=torch.float32, device=dev)
            nparts = ops.emb_fm_bwd_partials(dd.n_max, D)
            sq = torch.empty(nparts, dtype=torch.float32, device=dev)
            if g_lr is not None:
                G1 = torch.empty(dd.n_max, 1, dtype=torch.float32, device=dev)
                sq1 = torch.empty(nparts, dtype=torch.float32, device=dev)
        dnum = torch.empty(plan.Fd, D, dtype=torch.float32, device=dev) if plan.Fd else None
        dnum1 = torch.empty(plan.Fd, 1, dtype=torch.float32, device=dev) \
            if (plan.Fd and g_lr is not None) else None
        dbias = torch.empty(1, dtype=torch.float32, device=dev) \
            if (ctx.has_bias and g_lr is not None) else None
        ws = _Workspace.get(dev, ops.emb_fm_bwd_workspace_floats(dd.n_max if plan.C else 0, D,
                                                                 plan.Fd), tag="emb_fm_bwd")
        ops.emb_fm_bwd(d_out, ctx.out, ctx.S, g_fm, g_lr, plan.col_out_off, plan.C, D, dd, G, sq,
                       G1, sq1, ctx.dense, plan.num_out_off, B, dnum, dnum1, dbias, ws)
        if plan.C and ctx.sx is not None:
            # sharded: G / G1 are this rank's per-unique-key sums; their owners reduce across ranks
            group.shard_send_grads(ctx.sx, G)
            if G1 is not None:
                lr_group.shard_send_grads(ctx.sx, G1)
        elif plan.C:
            group.pending.append(_PendingGrad(dd, G, sq))
            if G1 is not None:
                lr_group.pending.append(_PendingGrad(dd, G1, sq1))
        if dnum is not None:
            group.add_num_grad(plan, dnum)
        if dnum1 is not None:
            lr_group.add_num_grad(lr_plan, dnum1)
        return (None, None, dbias) + (None,) * 10


class _SplitRecordFn(torch.autograd.Function):
    """The per-feature views of a gather record [B, n_slots, D] as ONE autograd node.  Plain
    slicing would give every view its own SliceBackward — a zero-filled record-sized gradient plus a
    copy per view, and an add per extra view — when a model reads the dict entry by entry (DIN:
    ~10 record-sized launches per step).  Here the backward concatenates the views' gradients
    (zeros for unused ones) into the record's gradient in one launch.  When the model takes the
    whole record instead (dict2tensor fast path) this node is never reached."""

    @staticmethod
    def forward(ctx, rec, bounds):
        ctx.shape, ctx.bounds = rec.shape, bounds
        return tuple(rec[:, lo, :] if w is None else rec[:, lo:lo + w, :] for lo, w in bounds)

    @staticmethod
    def backward(ctx, *grads):
        B, _, D = ctx.shape
        parts, like = [], next(g for g in grads if g is not None)
        for (lo, w), g in zip(ctx.bounds, grads):
            n = 1 if w is None else w
            if g is None:
                if parts and isinstance(parts[-1], int):
                    parts[-1] += n                  # one zero block per run of unused views
                else:
                    parts.append(n)
            else:
                parts.append(g.reshape(B, n, D))
        parts = [like.new_zeros(B, p, D) if isinstance(p, int) else p for p in parts]
        return torch.cat(parts, dim=1), None


class FeatureEmbeddingDict(nn.Module):
    """Native drop-in for feature_embedding.py:91-297 (same ctor / forward / dict2tensor)."""

    def __init__(self,
                 feature_map,
                 embedding_dim,
                 embedding_initializer="partial(nn.init.normal_, std=1e-4)",
                 required_feature_columns=None,
                 not_required_feature_columns=None,
                 use_pretrain=True,
                 use_sharing=True):
        super(FeatureEmbeddingDict, self).__init__()
        self._feature_map = feature_map
        self.required_feature_columns = required_feature_columns
        self.not_required_feature_columns = not_required_feature_columns
        self.use_pretrain = use_pretrain
        self.embedding_initializer = get_initializer(embedding_initializer)
        self.embedding_layers = nn.ModuleDict()
        self.feature_encoders = nn.ModuleDict()
        self._device = _alloc_device()
        self._groups = OrderedDict()   # D -> _TableGroup
        self._feat_group = {}          # feature -> D
        self._pooled_holes = set()     # raw sequence features with a reserved slot in front (DIN)
        self._tail_slots = 0           # reserved slots behind the last feature (DLRM)
        self._torch_feats = set()      # features served by stock torch modules ("embedding" type)
        self._stock_feats = set()      # id features delegated to the reference's PretrainedEmbedding
        lr_mode = (not (use_pretrain and use_sharing)) and embedding_dim == 1
        for feature, spec in self._feature_map.features.items():
            if not self.is_required(feature):
                continue
            ftype = spec["type"]
            if lr_mode:
                feat_dim = 1  # the LR trick, feature_embedding.py:135-138
                if ftype == "sequence":
                    self.feature_encoders[feature] = MaskedSumPooling()
            else:
                feat_dim = spec.get("embedding_dim", embedding_dim)
                if spec.get("feature_encoder", None):
                    self.feature_encoders[feature] = self.get_feature_encoder(spec["feature_encoder"])
                elif ftype == "embedding":
                    pretrain_dim = spec.get("pretrain_dim", feat_dim)
                    self.feature_encoders[feature] = nn.Linear(pretrain_dim, feat_dim, bias=False)
            share = spec.get("share_embedding")
            if ftype in ("categorical", "sequence") and use_sharing and share in self._stock_feats:
                # shares the table of a delegated feature: the same stock module serves both
                self.embedding_layers[feature] = self.embedding_layers[share]
                self._stock_feats.add(feature)
                continue
            if ftype in ("categorical", "sequence") and use_pretrain and "pretrained_emb" in spec \
                    and not (use_sharing and share in self._feat_group):
                # pretrained tables are not part of the native hot path: the
                # reference's own module is instantiated and called, exactly as
                # feature_embedding.py:156-171 does; its parameters are ordinary dense parameters
                self.embedding_layers[feature] = self._stock_pretrained(feature, spec, feat_dim,
                                                                        embedding_initializer)
                self._stock_feats.add(feature)
                continue
            width = spec["max_len"] if ftype == "sequence" else 1
            if ftype in ("categorical", "sequence", "numeric"):
                grp = self._groups.get(feat_dim)
                if grp is None:
                    grp = self._groups[feat_dim] = _TableGroup(feat_dim, self._device)
                    grp.owner = weakref.ref(self)
                self._feat_group[feature] = feat_dim
            if use_sharing and share in self._feat_group and ftype in ("categorical", "sequence"):
                if self._feat_group[share] != feat_dim:
                    raise NotImplementedError("share_embedding across different embedding dims")
                grp.add_alias(feature, share, width)
                continue
            if ftype == "numeric":
                grp.add_numeric(feature)
            elif ftype in ("categorical", "sequence"):
                grp.add_table(feature, spec["vocab_size"], spec.get("padding_idx", None), width)
            elif ftype == "embedding":
                self.embedding_layers[feature] = nn.Identity()
                self._torch_feats.add(feature)
            else:
                raise NotImplementedError("feature type={} is not supported.".format(ftype))
        for grp in self._groups.values():
            grp.allocate()
        self._bind_views()
        self._default_init()
        self.init_weights()

    # -- storage ----------------------------------------------------------------------------
    def _bind_views(self):
        """(Re)point the per-feature Parameters at views of the packed storage."""
        for grp in self._groups.values():
            for feature, (base, V, pidx) in grp.tables.items():
                lo, hi = grp.local_range(feature)
                view = grp.table[lo:hi]
                if feature in self.embedding_layers:
                    self.embedding_layers[feature].weight.data = view
                else:
                    self.embedding_layers[feature] = _TableView(view, pidx)
            for feature, owner in grp.alias.items():
                self.embedding_layers[feature] = self.embedding_layers[owner]
            for j, feature in enumerate(grp.numeric):
                view = grp.num_w[j].view(grp.D, 1)
                if feature in self.embedding_layers:
                    self.embedding_layers[feature].weight.data = view
                else:
                    self.embedding_layers[feature] = _NumericView(view)

    def _apply(self, fn, recurse=True):
        # nn.Module.to()/cuda()/float(): move the packed storages, then re-bind the views so the
        # per-feature Parameters keep aliasing one table (a per-Parameter move would split it).
        for grp in self._groups.values():
            names = ("table", "num_w", "m", "v", "last_step", "scal")
            if grp.record is not None:
                rec = fn(grp.record)
                if rec.dtype == torch.float32:
                    grp.owner = None          # (the views are re-bound once, below)
                    grp._set_record(rec)
                    grp.owner = weakref.ref(self)
                    names = ("num_w", "scal")
                else:
                    grp.owner = None
                    grp.drop_record()
                    grp.owner = weakref.ref(self)
            for name in names:
                t = getattr(grp, name)
                if t is not None:
                    setattr(grp, name, fn(t))
            ref = grp.table if grp.table is not None else grp.num_w
            if ref is not None and ref.device != grp.device:
                grp.device = ref.device
                grp.plans = {}
                grp.dedup_ws = None
        devs = [g.device for g in self._groups.values()]
        if devs:
            self._device = devs[0]
        self._bind_views()
        for module in self.feature_encoders.values():
            module._apply(fn)
        for f in self._torch_feats | self._stock_feats:
            self.embedding_layers[f]._apply(fn)
        return self

    def _stock_pretrained(self, feature, spec, feat_dim, embedding_initializer):
        """The reference's PretrainedEmbedding for one feature (pretrained_embedding.py:30-189),
        built with the arguments feature_embedding.py:157-171 passes."""
        try:
            from fuxictr.pytorch.layers.embeddings.pretrained_embedding import PretrainedEmbedding
        except ImportError as exc:
            raise NotImplementedError(
                "feature '%s' has `pretrained_emb`: it is delegated to the reference's "
                "PretrainedEmbedding module, which needs the `fuxictr` package importable (%s); "
                "drop `pretrained_emb` or pass use_pretrain=False" % (feature, exc))
        fmap = self._feature_map
        return PretrainedEmbedding(feature, spec,
                                   os.path.join(fmap.data_dir, spec["pretrained_emb"]),
                                   os.path.join(fmap.data_dir, "feature_vocab.json"),
                                   feat_dim, spec.get("pretrain_dim", feat_dim),
                                   spec.get("pretrain_usage", "init"), embedding_initializer)

    def _default_init(self):
        """What the stock modules would hold before init_weights(): nn.Embedding ~ N(0,1) with a
        zero padding row, nn.Linear(1,D) ~ kaiming-uniform (both are overwritten right after by
        init_weights / BaseModel.reset_parameters, as in the reference)."""
        with torch.no_grad():
            for grp in self._groups.values():
                if grp.table is not None:
                    grp.table.normal_(0.0, 1.0)
                    for _, (base, V, pidx) in grp.tables.items():
                        if pidx is not None and grp.local_row(base + pidx) is not None:
                            grp.table[grp.local_row(base + pidx)].zero_()
                    if grp.sharded:
                        grp.table[grp.rows_per_shard].zero_()   # the all-to-all pad row
                if grp.num_w is not None:
                    grp.num_w.uniform_(-1.0, 1.0)

    def get_feature_encoder(self, encoder):
        from . import layers  # noqa: F401  (encoder strings say "layers.MaskedAveragePooling()")
        try:
            if isinstance(encoder, list):
                return nn.Sequential(*[eval(enc) for enc in encoder])
            return eval(encoder)
        except Exception:
            raise ValueError("feature_encoder={} is not supported.".format(encoder))

    def init_weights(self):
        """feature_embedding.py:205-216 — initializer on rows 1.. when a padding_idx exists."""
        with torch.no_grad():
            for k, v in self.embedding_layers.items():
                if "share_embedding" in self._feature_map.features[k]:
                    continue
                if k in self._stock_feats:
                    v.init_weights()                      # feature_embedding.py:210-211
                    continue
                if isinstance(v, _TableView):
                    grp = self._groups[self._feat_group[k]]
                    if grp.sharded:
                        # a shard holds every n-th row: initialise the local rows, then restore
                        # the zero padding row if this rank owns it
                        if v.weight.numel():
                            self.embedding_initializer(v.weight)
                        base, _, pidx = grp.table_of(k)
                        if pidx is not None and grp.local_row(base + pidx) is not None:
                            grp.table[grp.local_row(base + pidx)].zero_()
                    elif v.padding_idx is not None:
                        self.embedding_initializer(v.weight[1:, :])
                    else:
                        self.embedding_initializer(v.weight)

    def is_required(self, feature):
        spec = self._feature_map.features[feature]
        if spec["type"] == "meta":
            return False
        elif self.required_feature_columns and (feature not in self.required_feature_columns):
            return False
        elif self.not_required_feature_columns and (feature in self.not_required_feature_columns):
            return False
        return True

    # -- forward ----------------------------------------------------------------------------
    def forward(self, inputs, feature_source=[], feature_type=[]):
        """-> OrderedDict name -> [B, D] (or [B, L, D]); values are views of one [B, F, D] record
        per embedding dim, written by a single gather launch."""
        fmap = self._feature_map.features
        present = []
        for feature in inputs.keys():
            spec = fmap.get(feature)
            if spec is None:
                continue
            if feature_source and not_in_whitelist(spec["source"], feature_source):
                continue
            if feature_type and not_in_whitelist(spec["type"], feature_type):
                continue
            if feature in self.embedding_layers:
                present.append(feature)
        present_set = set(present)
        emb, fused = {}, set()
        for D, grp in self._groups.items():
            feats = [f for f in fmap if f in present_set and self._feat_group.get(f) == D]
            if not feats:
                continue
            plan = grp.plan_for(feats, self._fused_pooling(grp, feats),
                                holes=tuple(f for f in feats if f in self._pooled_holes),
                                tail_slots=self._tail_slots if len(self._groups) == 1 else 0)
            ids, dense = grp.pack_inputs(plan, inputs)
            track = torch.is_grad_enabled() and self.training
            anchor = self._anchor(grp)
            B_ = (ids if ids is not None else dense).shape[0]
            front = None
            if self.fuse_front and grp.fused_front(plan) and len(self._groups) == 1:
                # the fused front end: gather (+ first-order term + FM term), one launch
                lr_mod, lr_grp, lr_plan = self._lr_peer_for(plan, feats, inputs)
                peers = (lr_grp,) if lr_grp is not None else ()
                dd = grp.prepare_train(plan, ids, inputs, peers) \
                    if (track and not grp.sharded) else None
                want_fm = bool(self._fuse_fm) and plan.n_slots == plan.C + plan.Fd \
                    and not (self._torch_feats or self._stock_feats)
                out, lr_out, fm_out, fm_lr = _EmbFMFn.apply(
                    anchor, lr_mod.embedding_layer.embedding_layer._anchor(lr_grp)
                    if lr_grp is not None else None,
                    lr_mod.bias if lr_grp is not None else None, grp, plan, lr_grp, lr_plan, ids,
                    dense, dd, inputs, want_fm, track)
                front = {"lr_mod": lr_mod, "lr": lr_out, "fm": fm_out, "fm_lr": fm_lr}
                if lr_grp is not None and hasattr(inputs, "cache"):
                    inputs.cache[("lr_out", id(lr_mod))] = lr_out
            else:
                if grp.table is not None and grp.table.dtype != torch.float32:
                    raise NotImplementedError(
                        "emb_dtype=bf16 is implemented for the fused column path only (every id "
                        "feature categorical with its own table, one embedding dim, batch <= 8192)")
                dd = grp.prepare_train(plan, ids, inputs) if (track and not grp.sharded) else None
                out = _EmbGatherFn.apply(anchor, grp, plan, ids, dense, dd, inputs, track)
            rec = out.view(out.shape[0], plan.n_slots, D)
            if front is not None:
                rec._fx_fused = front
            fused.update(plan.pooled)
            raw_seq = [fmap[f]["type"] == "sequence" and f not in plan.pooled for f in feats]
            # views in slot order; a reserved slot is one more (unnamed) view so that they tile the record
            entries = [(plan.slot[f][0], plan.slot[f][1] if r else None, f)
                       for f, r in zip(feats, raw_seq)] + [(h, None, None) for h in plan.hole.values()]
            if plan.tail0 is not None:
                entries += [(k, None, None) for k in range(plan.tail0, plan.n_slots)]
            entries.sort(key=lambda t: t[0])
            bounds = tuple((lo, w) for lo, w, _ in entries)
            if rec.requires_grad:
                views = _SplitRecordFn.apply(rec, bounds)
            else:
                views = _SplitRecordFn.forward(_NoCtx(), rec, bounds)
            emb.update((f, v) for (_, _, f), v in zip(entries, views) if f is not None)
            emb[("__record__", D)] = (rec, plan)
        for f in present:
            if f in self._torch_feats:
                emb[f] = self.embedding_layers[f](inputs[f].float())
            elif f in self._stock_feats:
                emb[f] = self.embedding_layers[f](inputs[f].long())   # feature_embedding.py:292-294
        feature_emb_dict = _EmbDict()
        for f in present:  # reference order = order of `inputs`
            e = emb[f]
            if f in self.feature_encoders and f not in fused:
                e = self.feature_encoders[f](e)
                feature_emb_dict._encoded.add(f)
            feature_emb_dict[f] = e
        feature_emb_dict._records = [v for k, v in emb.items() if isinstance(k, tuple)]
        feature_emb_dict._orig = {f: id(t) for f, t in feature_emb_dict.items()}
        if self._pooled_holes:
            # a linked DIN_Attention (link_fusion) finds the record and the packed ids of THIS forward here
            self.__dict__["_fx_last"] = (inputs, feature_emb_dict)
        return feature_emb_dict

    def reserve_pooled_slot(self, feature):
        """Native extension: keep one record slot free in front of the positions of the raw sequence
        `feature`; a model that replaces the sequence by a pooled vector (DIN) writes it there."""
        self._pooled_holes.add(feature)

    def reserve_tail_slots(self, n):
        """Native extension: n record slots behind the last feature that the gather does not write (a
        model appends its own vectors to the field list there: DLRM's bottom-tower output)."""
        self._tail_slots = int(n)

    fuse_pooling = True     # class switch for A/B measurements (scripts/seqpool_bench.py)
    fuse_front = True       # class switch: the fused front / back end of csrc/fx_fused.hip
    _lr_peer = None         # the model's LogisticRegression, set by link_fusion()
    _fuse_fm = False        # the model contains a product_sum interaction over this layer's record

    def _lr_peer_for(self, plan, feats, inputs):
        """-> (LogisticRegression module, its D=1 table group, its plan) when the model's first-order
        layer looks up the SAME id columns / numeric columns as `plan` (then both share one de-dup
        and one launch), else (None, None, None)."""
        lr = self._lr_peer
        if lr is None or not hasattr(inputs, "cache"):
            return None, None, None
        layer = lr.embedding_layer.embedding_layer
        groups = layer.table_groups()
        fmap = layer._feature_map.features
        lr_feats = [f for f in fmap if f in inputs and f in layer.embedding_layers]
        if (len(groups) != 1 or layer._torch_feats or layer._stock_feats or lr_feats != list(feats)
                or groups[0].D != 1
                or groups[0].sharded != next(iter(self._groups.values())).sharded
                or lr.training != self.training):
            return None, None, None
        lr_grp = groups[0]
        lr_plan = lr_grp.plan_for(lr_feats, tail=[f for f in lr_feats
                                                   if fmap[f]["type"] == "sequence"])
        if lr_plan.sig != plan.sig or lr_plan.pack_sig != plan.pack_sig \
                or lr_plan.num_feats != plan.num_feats:
            return None, None, None
        return lr, lr_grp, lr_plan

    def _fused_pooling(self, grp, feats):
        """Sequence features of `feats` whose encoder is one of the two pooling layers: reduced
        inside the gather instead of materialising [B, L, D] for torch.sum."""
        if not self.fuse_pooling or _lib.row_lanes(grp.D) > 64:
            return None
        modes = {}
        for f in feats:
            enc = self.feature_encoders[f] if f in self.feature_encoders else None
            if self._feature_map.features[f]["type"] != "sequence" or f not in grp.widths:
                continue
            if type(enc) is MaskedSumPooling:
                modes[f] = ops.POOL_SUM
            elif type(enc) is MaskedAveragePooling:
                modes[f] = ops.POOL_MEAN
        return modes or None

    def _anchor(self, grp):
        for f in grp.tables:
            return self.embedding_layers[f].weight
        return self.embedding_layers[grp.numeric[0]].weight

    @staticmethod
    def packed_ids(inputs, feature):
        """int32 view [B, width] of `feature`'s id columns in the packed id matrix of this batch (the
        forward packs it once, FeatureDict.cache), or None.  Lets a model use the raw ids as a
        padding mask (`X[f].long() != 0`, DIN.py:125) without cast / compare launches."""
        cache = getattr(inputs, "cache", None) or {}
        for key, val in cache.items():
            if isinstance(key, tuple) and key and key[0] == "pack" and val[0] is not None:
                col = 0
                for name, width in key[1][0]:
                    if name == feature:
                        return val[0][:, col:col + width]
                    col += width
        return None

    def dict2tensor(self, embedding_dict, flatten_emb=False, feature_list=[], feature_source=[],
                    feature_type=[]):
        """feature_embedding.py:230-259.  When the selection is a contiguous slot range of the
        gather record the record itself is returned (no stack/cat pass)."""
        names = []
        for feature, spec in self._feature_map.features.items():
            if feature_list and not_in_whitelist(feature, feature_list):
                continue
            if feature_source and not_in_whitelist(spec["source"], feature_source):
                continue
            if feature_type and not_in_whitelist(spec["type"], feature_type):
                continue
            if feature in embedding_dict:
                names.append(feature)
        handed = getattr(embedding_dict, "_fx_flat", None)
        if handed is not None and flatten_emb and not (feature_list or feature_source or feature_type):
            # the attention already ran inside the gather record (DIN_Attention._try_in_record): the
            # tower's input [fields.., attended vector] IS the record's prefix, provided the caller put
            # exactly that attended vector into the dict in place of the sequence (DIN.py:127-130)
            flat, seq, hole, D = handed
            t = embedding_dict.get(seq)
            same = (t is not None and t.dim() == 2 and t.shape[1] == D and t.stride(-1) == 1
                    and t.data_ptr() == flat.data_ptr() + hole * D * flat.element_size()
                    and names and names[-1] == seq and len(names) == hole + 1
                    and all(embedding_dict._orig.get(f) == id(embedding_dict[f]) for f in names[:-1]))
            if same:
                return flat
        fast = self._record_slice(embedding_dict, names)
        if fast is not None:
            return fast.flatten(start_dim=1) if flatten_emb else fast
        if flatten_emb:
            return torch.cat(self._merged_runs(embedding_dict, names), dim=-1)
        return torch.stack([embedding_dict[f] for f in names], dim=1)

    @staticmethod
    def _merged_runs(embedding_dict, names):
        """Pieces to concatenate for flatten_emb.  Untouched entries are views handed out by ONE
        autograd node (_SplitRecordFn), so concatenating them feature by feature costs one launch
        forward and one backward, however many there are."""
        return [embedding_dict[f] for f in names]

    @staticmethod
    def _record_slice(embedding_dict, names):
        records = getattr(embedding_dict, "_records", None)
        if not records or len(records) != 1 or not names:
            return None
        rec, plan = records[0]
        lo = None
        nxt = None
        for f in names:
            if f not in plan.slot or f in embedding_dict._encoded:
                return None
            if embedding_dict._orig.get(f) != id(embedding_dict[f]):
                return None          # the caller replaced this entry (e.g. DIN's pooled sequence)
            s, w = plan.slot[f]
            if w != 1:
                return None
            if lo is None:
                lo = s
            elif s != nxt:
                return None
            nxt = s + 1
        if lo == 0 and nxt == plan.n_slots:
            return rec
        return rec[:, lo:nxt, :]

    # -- sharded checkpoints ------------------------------------------------------------------
    def load_full_tables(self, full_state, prefix=""):
        """Copy this rank's rows out of a FULL (unsharded, reference-layout) state dict."""
        with torch.no_grad():
            for grp in self._groups.values():
                for feature, (base, V, _) in grp.tables.items():
                    w = full_state[prefix + "embedding_layers." + feature + ".weight"]
                    lo, hi = grp.local_range(feature)
                    if hi > lo:
                        first = lo * grp.n_shards + grp.rank - base      # row inside the feature
                        grp.table[lo:hi] = w[first::grp.n_shards].to(grp.device)
                for j, feature in enumerate(grp.numeric):
                    w = full_state[prefix + "embedding_layers." + feature + ".weight"]
                    grp.num_w[j] = w.reshape(-1).to(grp.device)

    def gather_full_tables(self, prefix=""):
        """All ranks -> a FULL state dict (reference layout) on every rank (tests, small tables)."""
        out = {}
        for grp in self._groups.values():
            for feature, (base, V, _) in grp.tables.items():
                full = torch.zeros(V, grp.D, dtype=torch.float32, device=grp.device)
                lo, hi = grp.local_range(feature)
                if hi > lo:
                    first = lo * grp.n_shards + grp.rank - base
                    full[first::grp.n_shards] = grp.table[lo:hi]
                if grp.dist is not None:
                    grp.dist.all_reduce_sum(full)
                out[prefix + "embedding_layers." + feature + ".weight"] = full
                for f2, owner in grp.alias.items():
                    if owner == feature:
                        out[prefix + "embedding_layers." + f2 + ".weight"] = full
            for j, feature in enumerate(grp.numeric):
                out[prefix + "embedding_layers." + feature + ".weight"] = \
                    grp.num_w[j].view(grp.D, 1).clone()
        return out

    # -- optimizer hooks --------------------------------------------------------------------
    def table_groups(self):
        for grp in self._groups.values():
            grp.owner = weakref.ref(self)     # (a deep copy of the module carries the original's weak reference)
        return list(self._groups.values())

    def table_parameters(self):
        """Parameters that are views of packed tables (handled by the sparse-row optimizer)."""
        seen, out = set(), []
        for m in self.embedding_layers.values():
            if isinstance(m, _TableView) and id(m.weight) not in seen:
                seen.add(id(m.weight))
                out.append(m.weight)
        return out

    def numeric_parameters(self):
        return [m.weight for m in self.embedding_layers.values() if isinstance(m, _NumericView)]


class _NoCtx(object):
    """Stand-in ctx for calling an autograd Function's forward outside autograd."""
    pass


class _EmbDict(OrderedDict):
    """OrderedDict of embeddings that remembers the gather record(s) it is a view of."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self._records = []
        self._encoded = set()
        self._orig = {}


class FeatureEmbedding(nn.Module):
    """feature_embedding.py:30-88."""

    def __init__(self,
                 feature_map,
                 embedding_dim,
                 embedding_initializer="partial(nn.init.normal_, std=1e-4)",
                 required_feature_columns=None,
                 not_required_feature_columns=None,
                 use_pretrain=True,
                 use_sharing=True):
        super(FeatureEmbedding, self).__init__()
        self.embedding_layer = FeatureEmbeddingDict(
            feature_map, embedding_dim, embedding_initializer=embedding_initializer,
            required_feature_columns=required_feature_columns,
            not_required_feature_columns=not_required_feature_columns,
            use_pretrain=use_pretrain, use_sharing=use_sharing)

    def forward(self, X, feature_source=[], feature_type=[], flatten_emb=False):
        feature_emb_dict = self.embedding_layer(X, feature_source=feature_source,
                                                feature_type=feature_type)
        out = self.embedding_layer.dict2tensor(feature_emb_dict, flatten_emb=flatten_emb)
        out._fx_records = getattr(feature_emb_dict, "_records", None)   # (record, plan) pairs behind it
        return out


# ------------------------------------------------------------------------------------------------
# LR / FM
# ------------------------------------------------------------------------------------------------
class _LRFn(torch.autograd.Function):
    """out[b] = sum of the D=1 rows + numeric terms + bias, one launch (logistic_regression.py:55-58)."""

    @staticmethod
    def forward(ctx, anchor, bias, group, plan, ids, dense, dd, inputs, track):
        B = (ids if ids is not None else dense).shape[0]
        out = torch.empty(B, 1, dtype=torch.float32, device=group.device)
        num_w1 = group.select_num_w(plan)
        sx = None
        if group.sharded and plan.C:
            sx = group.shard_exchange_ids(plan, ids, inputs, track)
            rows = group.shard_fetch_rows(sx, track).contiguous()
            ops.lr_fwd(rows, sx.lookup_slot, sx.slot_base, sx.slot_vocab, dense, num_w1, bias,
                       out, group.ensure_scal())
        else:
            ops.lr_fwd(group.table, ids, plan.col_row_base, plan.col_vocab, dense, num_w1, bias,
                       out, group.ensure_scal())
        ctx.group, ctx.plan, ctx.ids, ctx.dense, ctx.dd, ctx.sx = group, plan, ids, dense, dd, sx
        ctx.inputs = inputs if hasattr(inputs, "cache") else None
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous()
        # every column of a sample receives the same upstream value: ld = 1, offsets = 0
        ctx.group.backward(ctx.plan, ctx.ids, ctx.dense, dout, 1, ctx.plan.col_zero_off,
                           ctx.plan.num_zero_off, ctx.dd, ctx.inputs, sx=ctx.sx)
        dbias = dout.sum().reshape(1) if ctx.has_bias else None
        return None, dbias, None, None, None, None, None, None, None


class LogisticRegression(nn.Module):
    """logistic_regression.py:24-59."""

    def __init__(self, feature_map, use_bias=True):
        super(LogisticRegression, self).__init__()
        self.bias = nn.Parameter(torch.zeros(1, device=_alloc_device()), requires_grad=True) \
            if use_bias else None
        self.embedding_layer = FeatureEmbedding(feature_map, 1, use_pretrain=False,
                                                use_sharing=False)

    def forward(self, X):
        cached = X.cache.pop(("lr_out", id(self)), None) if hasattr(X, "cache") else None
        if cached is not None:
            return cached        # computed by the embedding layer's fused launch (_EmbFMFn)
        layer = self.embedding_layer.embedding_layer
        groups = layer.table_groups()
        fmap = layer._feature_map.features
        feats = [f for f in fmap if f in X and f in layer.embedding_layers]
        native = (len(groups) == 1 and not layer._torch_feats
                  and all(f in layer._feat_group for f in feats))
        if not native:
            embed_weights = self.embedding_layer(X)
            output = embed_weights.sum(dim=1)
            if self.bias is not None:
                output = output + self.bias
            return output
        grp = groups[0]
        # lr_fwd sums every column (= the sum pooling this layer installs for sequences); their
        # columns go last like in the embedding layer's pooled plan, so both share one de-dup
        plan = grp.plan_for(feats, tail=[f for f in feats if fmap[f]["type"] == "sequence"])
        ids, dense = grp.pack_inputs(plan, X)
        track = torch.is_grad_enabled() and self.training
        dd = grp.prepare_train(plan, ids, X) if (track and not grp.sharded) else None
        return _LRFn.apply(layer._anchor(grp), self.bias, grp, plan, ids, dense, dd, X, track)


class _FMFn(torch.autograd.Function):
    """0.5 * sum_d((sum_f e)^2 - sum_f e^2) (+ addend), inner_product.py:55-62."""

    @staticmethod
    def forward(ctx, emb, addend):
        emb = emb.contiguous()
        B, F, D = emb.shape
        out = torch.empty(B, 1, dtype=torch.float32, device=emb.device)
        ops.fm_fwd(emb.view(B, F * D), F, D, addend, out)
        ctx.save_for_backward(emb)
        ctx.has_add = addend is not None
        return out

    @staticmethod
    def backward(ctx, g):
        (emb,) = ctx.saved_tensors
        B, F, D = emb.shape
        g = g.contiguous()
        demb = torch.empty_like(emb)
        ops.fm_bwd(emb.view(B, F * D), F, D, g, demb.view(B, F * D), accumulate=False)
        return demb, (g if ctx.has_add else None)


class _DotInteractFn(torch.autograd.Function):
    """All pairwise field dots (DLRM "dot"), inner_product.py:63-66."""

    @staticmethod
    def forward(ctx, emb):
        emb = emb.contiguous()
        B, F, D = emb.shape
        out = torch.empty(B, F * (F - 1) // 2, dtype=torch.float32, device=emb.device)
        ops.dot_interact_fwd(emb.view(B, F * D), F, D, out)
        ctx.save_for_backward(emb)
        return out

    @staticmethod
    def backward(ctx, g):
        (emb,) = ctx.saved_tensors
        B, F, D = emb.shape
        demb = torch.empty_like(emb)
        ops.dot_interact_bwd(emb.view(B, F * D), g.contiguous(), F, D, demb.view(B, F * D))
        return demb


class _DlrmMixFn(torch.autograd.Function):
    """DLRM's `dot` interaction with the bottom tower's vector as the LAST field of the gather record
    (native fast path of zoo.DLRM, DLRM.py:110-121): rec [B, F, D] = [embeddings.. | dense vector], the
    dense vector having been written into the record's reserved slot by the bottom tower's last GEMM.
      forward : ONE launch -> [pairwise dots | dense vector | zero padding] = the top tower's (aligned)
                input: no cat of the fields, no cat of the products with the dense vector, no padding cat;
      backward: ONE launch -> the record's gradient, the dense vector's direct share added to its slot;
                the slot's view is the bottom tower's gradient, the rest the embedding backward's.
    `dense_vec` is an argument only so that its gradient has somewhere to go (its values are in rec)."""

    @staticmethod
    def forward(ctx, rec, dense_vec, pad):
        B, F, D = rec.shape
        P = F * (F - 1) // 2
        out = torch.empty(B, P + D + pad, dtype=torch.float32, device=rec.device)
        ops.dot_interact_fwd(rec.view(B, F * D), F, D, out, tail=D + pad)
        ctx.save_for_backward(rec)
        ctx.pad = pad
        return out

    @staticmethod
    def backward(ctx, g):
        (rec,) = ctx.saved_tensors
        B, F, D = rec.shape
        if g.stride(-1) != 1:
            g = g.contiguous()
        demb = torch.empty(B, F, D, dtype=torch.float32, device=rec.device)
        ops.dot_interact_bwd(rec.view(B, F * D), g, F, D, demb.view(B, F * D), tail=D + ctx.pad)
        # (the dense vector's gradient is a COPY of its slot, B x D floats: handing out a view of demb
        # would let autograd's in-place gradient accumulation corrupt it once rec gains a second
        # consumer — ADVICE r3)
        return demb, demb[:, F - 1, :].clone(), None


class InnerProductInteraction(nn.Module):
    """inner_product.py:23-70.  `product_sum` is native; the other outputs run the reference's
    torch formulas (not on the BASELINE path)."""

    def __init__(self, num_fields, output="product_sum"):
        super(InnerProductInteraction, self).__init__()
        kinds = ("product_sum", "bi_interaction", "inner_product", "elementwise_product")
        if output not in kinds:
            raise ValueError("InnerProductInteraction output={} is not supported.".format(output))
        self._output_type = output
        dev = _alloc_device()
        # buffers the reference registers as frozen Parameters (same names -> same state_dict keys)
        if output == "inner_product":
            self.interaction_units = num_fields * (num_fields - 1) // 2
            upper = torch.ones(num_fields, num_fields, device=dev).triu(1).bool()
            self.triu_mask = nn.Parameter(upper, requires_grad=False)
        elif output == "elementwise_product":
            pairs = torch.triu_indices(num_fields, num_fields, offset=1).to(dev)
            self.triu_index = nn.Parameter(pairs, requires_grad=False)

    def forward(self, feature_emb):
        kind = self._output_type
        if kind == "product_sum":
            fused = getattr(feature_emb, "_fx_fused", None)
            if fused is not None and fused["fm"] is not None:
                return fused["fm"]       # computed together with the gather (_EmbFMFn)
            return _FMFn.apply(feature_emb, None)
        if kind == "inner_product":
            n_f, dim = feature_emb.shape[1], feature_emb.shape[2]
            if n_f * dim <= 4096 and n_f * (n_f - 1) // 2 <= 4096:
                return _DotInteractFn.apply(feature_emb)
            gram = torch.bmm(feature_emb, feature_emb.transpose(1, 2))      # beyond the LDS tile
            return gram.masked_select(self.triu_mask).view(-1, self.interaction_units)
        if kind == "bi_interaction":      # 0.5 * ((sum_f e)^2 - sum_f e^2), kept per dimension
            s1 = feature_emb.sum(dim=1)
            return 0.5 * (s1 * s1 - (feature_emb * feature_emb).sum(dim=1))
        left, right = (feature_emb.index_select(1, idx) for idx in self.triu_index)
        return left * right


class FactorizationMachine(nn.Module):
    """factorization_machine.py:25-59; fm + lr fused into the FM kernel's addend."""

    def __init__(self, feature_map):
        super(FactorizationMachine, self).__init__()
        self.fm_layer = InnerProductInteraction(feature_map.num_fields, output="product_sum")
        self.lr_layer = LogisticRegression(feature_map, use_bias=True)

    def forward(self, X, feature_emb):
        fused = getattr(feature_emb, "_fx_fused", None)
        if fused is not None and fused["lr_mod"] is self.lr_layer and fused["fm_lr"] is not None:
            if hasattr(X, "cache"):
                X.cache.pop(("lr_out", id(self.lr_layer)), None)
            return fused["fm_lr"]        # gather + LR + FM came out of ONE launch (_EmbFMFn)
        lr_out = self.lr_layer(X)
        return _FMFn.apply(feature_emb, lr_out)


# ------------------------------------------------------------------------------------------------
# dense tower
# ------------------------------------------------------------------------------------------------
import os as _os
_FORCE_SPLITK = int(_os.environ.get("FX_DW_SPLITK", "0"))
_MLP_PAD = _os.environ.get("FX_MLP_PAD", "1") != "0"      # A/B switch of the 4-float input padding


def _split_k_for(M, N, K):
    """Split the contraction when the output grid alone cannot fill 256 CUs (weight gradients): aim
    for ~1024 workgroups of 64x64 — the 4 per CU the 64x64 kernel keeps resident (measured,
    profiles/r02_gemm_probe.txt, GEMM + slab reduce: 1024x1024x4096 85.2 us at 2 splits, 80.9 at 4;
    1024x624x4096 82.0 / 64.6; 624x624x4096 62.9 / 50.6 / 48.8 at 2 / 4 / 8)."""
    if M <= 4:                      # skinny weight gradient: column-parallel reduction kernel
        # enough K slabs to put >= 4 workgroups on every CU (N/256 column blocks x slabs); the wide
        # slab reduce handles hundreds of slabs in one small launch
        return max(1, min(512 if N <= 256 else 256, K // 16))
    tiles = ((M + 63) // 64) * ((N + 63) // 64)
    if _FORCE_SPLITK:                  # FX_DW_SPLITK=<n>: experiment switch
        return max(1, min(_FORCE_SPLITK, K // 256))
    if tiles >= 768:
        return 1
    if tiles <= 4:
        return max(1, min(-(-512 // tiles), K // 256, 256))
    want = 1024.0 / tiles
    s = 1
    while s * 1.5 < want:          # nearest power of two (log scale)
        s *= 2
    return max(1, min(s, K // 256, 16))


class _Workspace(object):
    """Grow-only fp32 scratch per device for split-K slabs and column-sum partials."""
    _bufs = {}

    @classmethod
    def get(cls, device, n, tag=None):
        buf = cls._bufs.get((device, tag))
        if buf is None or buf.numel() < n:
            buf = torch.empty(max(n, 1 << 20 if tag is None else 1), dtype=torch.float32,
                              device=device)
            cls._bufs[(device, tag)] = buf
        return buf


def linear_grads(dz, x, W, need_bias, mask=None, add=None, dx_out=None):
    """dW, db (as linear_weight_grads) and dx = dz W (+ the ReLU `mask` of the layer below / the
    residual `add` in the epilogue), the two products in one launch.  dx_out: where dx goes (a
    row-strided [B, K_in] view of a wider buffer is written in place)."""
    Bsz, N_out = dz.shape
    K_in = x.shape[1]
    dW = torch.empty(N_out, K_in, dtype=torch.float32, device=dz.device)
    dx = dx_out if dx_out is not None else \
        torch.empty(Bsz, K_in, dtype=torch.float32, device=dz.device)
    # split_k is the LARGEST slab count the workspace holds: the library picks the actual K split of
    # the launch it builds (fx_gemm_f32_batch: one grid for both products, tiles and slabs chosen
    # together), never more than this
    sk = _split_k_for(N_out, K_in, Bsz)
    if N_out > 4 and K_in > 8 and not _FORCE_SPLITK:
        sk = max(sk, min(8, Bsz // 256))
    sk = max(sk, 1)
    ws = _Workspace.get(dz.device, ops.gemm_workspace_floats(N_out, K_in, sk))
    db = torch.empty(N_out, dtype=torch.float32, device=dz.device) if need_bias else None
    ops.gemm_dw_dx(dz, x, W, dW, dx, split_k=sk, workspace=ws, rowsum=db, mask=mask, add=add)
    return dW, db, dx


def linear_weight_grads(dz, x, W_shape, need_bias):
    """dW[N_out, K_in] = dz^T x (split-K), db[N_out] = colsum(dz)."""
    Bsz, N_out = dz.shape
    K_in = x.shape[1]
    dW = torch.empty(N_out, K_in, dtype=torch.float32, device=dz.device)
    sk = _split_k_for(N_out, K_in, Bsz)
    ws = _Workspace.get(dz.device, ops.gemm_workspace_floats(N_out, K_in, sk))
    db = torch.empty(N_out, dtype=torch.float32, device=dz.device) if need_bias else None
    # the bias gradient (column sums of dz) rides along in the dW GEMM: its A tiles ARE dz
    ops.gemm(dz, x, dW, transa=True, transb=False, split_k=sk, workspace=ws, rowsum=db)
    return dW, db


_ZERO_COLS = {}


def _zero_cols(rows, cols, device):
    key = (rows, cols, device)
    z = _ZERO_COLS.get(key)
    if z is None:
        z = _ZERO_COLS[key] = torch.zeros(rows, cols, dtype=torch.float32, device=device)
    return z


class _HeadCtx(object):
    """The labels of ONE training step, announced by BaseModel._forward_backward before the forward pass
    (rank_model.py:307-323 runs forward -> compute_loss -> backward; nothing in between needs the loss
    value): a tower that ends in Linear(K -> 1) can then run its head forward, the sigmoid + BCE and the
    head's backward in one pass over its top hidden layer (ops.head_train).  Whether that output really
    was the model's logit is only known later: the loss (rank_model._bce_loss) takes the fused result if
    it is handed exactly this logit tensor, the tower's backward takes the fused gradients if the gradient
    it receives is exactly the fused dlogit — anything else (the logit went through more arithmetic: the
    reference's `y_pred += mlp(...)`, DeepFM.py:86-87) falls back to the separate kernels, and the module
    stops asking (`_fx_head_off`)."""
    __slots__ = ("_y", "_labels", "root_scale", "root_ptr", "result")

    def __init__(self, labels, root_scale, root_ptr):
        # `labels`: a callable, asked at the first offer — i.e. inside the towers, AFTER the model's
        # get_inputs() staged this batch (BaseModel.get_labels reads the labels out of that staging buffer;
        # before it they are the previous batch's)
        self._labels, self._y = labels, None
        self.root_scale, self.root_ptr = root_scale, root_ptr
        self.result = None            # (logit, dlogit, loss, logit._version) of the last head that took the offer

    @property
    def y(self):
        if self._y is None:
            y = self._labels()
            self._y = y if y.is_contiguous() else y.contiguous()
        return self._y

    @property
    def labels_taken(self):
        return self._y


_HEAD_CTX = None
_HEAD_FUSED = os.environ.get("FX_HEAD_FUSED", "1") != "0"
A2A_FILL_PROBE = {"on": False, "max_used": 0, "cap": 0, "even_share": 0}     # see _TableGroup.shard_exchange_ids


class _ReluNote(object):
    """Left by a node whose output buffer is [plain | ReLU'd] columns (_CrossDeepFn: [cross | deep]) for the
    head that reads that buffer: the fused head applies the ReLU mask of the columns >= `col` to the gradient
    it hands back (`masked_ptr`: that gradient tensor, set when it really is used); the node skips its own
    mask launch if the gradient it receives is that very tensor — a sum with some other consumer's gradient
    is a new tensor and is masked as always (the mask is idempotent)."""
    __slots__ = ("col", "masked_ptr")

    def __init__(self, col):
        self.col, self.masked_ptr = col, None


_RELU_NOTES = {}      # data_ptr of such a buffer -> _ReluNote, for the duration of one step's forward


def _head_offer(module, rows, out_features):
    """The step's _HeadCtx if `module` (a tower ending in `out_features` == 1 over `rows` samples) may use it."""
    hc = _HEAD_CTX
    if hc is None or out_features != 1 or rows != hc.y.numel() or module.__dict__.get("_fx_head_off"):
        return None
    return hc, module, None


class _MLPFn(torch.autograd.Function):
    """Whole Linear(+ReLU) stack as ONE autograd node: forward = one GEMM per layer with
    bias+ReLU in the epilogue; backward = dX GEMM with the ReLU mask of the layer below in its
    epilogue, split-K dW GEMM, column-sum db.  args = (x, acts, W0, b0, W1, b1, ...)."""

    @staticmethod
    def forward(ctx, x, acts, out_add, dx_into, out_into, head, *wb):
        """head: None or _head_offer()'s (step context, asking module) — see _HeadCtx.
        out_add: optional [B, N_last] tensor added to the last layer's output in its epilogue (DeepFM:
        logit = fm + mlp, DeepFM.py:87 — one ATen add launch less); its gradient is dy.
        dx_into: optional callable -> a [B, K0] tensor (unit inner stride) the input's gradient is
        written into (DIN: the head of the gather record's gradient).
        out_into: optional [B, N_last] tensor (unit inner stride) the LAST layer writes its result into and
        that is returned (DLRM: a slot of the gather record).
        An input that is already zero-padded to the aligned width (K0 + pad columns) is taken as it is."""
        need_dx = x.requires_grad
        if x.stride(-1) != 1 or x.stride(0) % 4 or x.data_ptr() % 16:
            x = x.contiguous()       # a 16-byte aligned row-strided view (record prefix) is read in place
        n = len(acts)
        # An input width that is not a multiple of 4 floats (DLRM's top MLP reads 27*26/2 + 16 = 367
        # columns) leaves the rows of x and W0 4-byte aligned: the GEMMs would take the unpipelined
        # kernel (55 us instead of 28 for 4096 x 1024 x 367, and no dW + dX pair).  One zero column
        # more on both operands changes no sum and keeps every row 16-byte aligned.
        K0 = wb[0].shape[1]
        pad = (-K0) % 4 if (K0 >= 64 and _MLP_PAD) else 0
        W0p = None
        pre_padded = pad > 0 and x.shape[1] == K0 + pad
        if pad:
            if not pre_padded:
                x = torch.cat([x, _zero_cols(x.shape[0], pad, x.device)], dim=1)      # one launch each
            W0p = torch.cat([wb[0], _zero_cols(wb[0].shape[0], pad, x.device)], dim=1)
        hs = [x]
        h = x
        ctx.head = None
        for i in range(n):
            W, b = wb[2 * i], wb[2 * i + 1]
            if i == 0 and W0p is not None:
                W = W0p
            if i == n - 1 and out_into is not None:
                y = out_into
            else:
                y = torch.empty(h.shape[0], W.shape[0], dtype=torch.float32, device=h.device)
            if (i == n - 1 and head is not None and not acts[i] and out_into is None
                    and not (n == 1 and (W0p is not None or dx_into is not None))
                    and ops.head_train_ok(h, W, out_add)):
                hc, asker, note = head
                rows, K = h.shape
                if n > 1:
                    mask_from, note = (0 if acts[n - 2] else -1), None
                else:
                    mask_from = note.col if (note is not None and note.col % 4 == 0) else -1
                    note = note if mask_from >= 0 else None
                dlogit = torch.empty(rows, 1, dtype=torch.float32, device=h.device)
                dzp = torch.empty(rows, K, dtype=torch.float32, device=h.device) \
                    if (n > 1 or need_dx) else None
                dW = torch.empty(1, K, dtype=torch.float32, device=h.device)
                db = torch.empty(1, dtype=torch.float32, device=h.device) if b is not None else None
                loss = torch.empty((), dtype=torch.float32, device=h.device)
                ws = _Workspace.get(h.device, ops.head_train_workspace_floats(rows, K), tag=("head", K))
                ops.head_train(h, W, b, out_add, hc.y, mask_from, hc.root_scale, y, dlogit, dzp, dW, db, loss, ws)
                # (the logit's version counter: a model that goes on to modify it IN PLACE — AutoInt.py:112-116
                # `y_pred += self.lr_layer(X)` — hands the loss the same storage with another value; views share
                # the counter with their base)
                # (a WEAK reference to the output: ctx -> y -> grad_fn -> ctx would be a cycle that only this
                # node's backward breaks — a head-fused forward that is never followed by backward would keep the
                # logit and the fused gradients alive until the cyclic GC runs; hc.result owns y)
                ctx.head = (dlogit, dzp, dW, db, asker, note, weakref.ref(y), y._version)
                hc.result = (y, dlogit, loss, y._version)
            else:
                ops.gemm(h, W, y, transa=False, transb=True, bias=b, act=1 if acts[i] else 0,
                         add=out_add if (i == n - 1 and out_add is not None) else None)
            # (the tower's own output is only needed by the backward as the ReLU mask of an activated last layer:
            # otherwise it is not kept on ctx — the same cycle)
            hs.append(y if (i < n - 1 or acts[i]) else None)
            h = y
        ctx.has_add = out_add is not None
        ctx.acts = acts
        ctx.wb = wb
        ctx.hs = hs
        ctx.need_dx = need_dx
        ctx.K0, ctx.W0p, ctx.pre_padded = K0, W0p, pre_padded
        ctx.dx_into = dx_into if (need_dx and W0p is None) else None
        return h

    @staticmethod
    def backward(ctx, dy):
        acts, wb, hs = ctx.acts, ctx.wb, ctx.hs
        n = len(acts)
        grads = [None] * (2 * n)
        dx = None
        top = n - 1
        head_used = False
        if ctx.head is not None:
            dlogit, dzp, dWh, dbh, asker, note, y_ref, y_ver = ctx.head
            y_head = y_ref()
            # (no second owner of dW / db: AccumulateGrad takes over a gradient it holds alone and
            # CLONES one somebody else still references — two copy launches per step)
            ctx.head = None
            if (dy.data_ptr() == dlogit.data_ptr() and dy.numel() == dlogit.numel()
                    and y_head is not None and y_head._version == y_ver):
                # the gradient that arrives IS the fused dlogit: the head's own gradients and the
                # gradient below it were formed in the forward pass (ops.head_train)
                head_used = True
                grads[2 * top], grads[2 * top + 1] = dWh, dbh
                if note is not None and dzp is not None:
                    note.masked_ptr = dzp.data_ptr()   # THIS tensor carries the producer's ReLU mask
                if n == 1:
                    return (dzp if ctx.need_dx else None, None, dy if ctx.has_add else None, None, None,
                            None) + tuple(grads)
                dz = dzp
                top = n - 2
            else:
                asker.__dict__["_fx_head_off"] = True        # this tower's output is not the logit
        if not head_used:
            if acts[n - 1]:
                # (a column slice of the gradient of the torch.cat that joins the towers is read in place)
                dz = ops.mask_mul(dy if dy.stride(-1) == 1 else dy.contiguous(), hs[n],
                                  torch.empty_like(hs[n]))
            else:
                dz = dy.contiguous()
        for i in range(top, -1, -1):
            W, b = wb[2 * i], wb[2 * i + 1]
            if i == 0 and ctx.W0p is not None:
                W = ctx.W0p
            h_in = hs[i]
            if i > 0 or ctx.need_dx:
                mask = h_in if (i > 0 and acts[i - 1]) else None
                dx_out = ctx.dx_into() if (i == 0 and ctx.dx_into is not None) else None
                dW, db, dh = linear_grads(dz, h_in, W, b is not None, mask=mask, dx_out=dx_out)
                if i > 0:
                    dz = dh
                else:
                    dx = dh
            else:
                dW, db = linear_weight_grads(dz, h_in, W.shape, b is not None)
            if i == 0 and ctx.W0p is not None:        # drop the zero column again (views)
                dW = dW[:, :ctx.K0]
                if dx is not None and not ctx.pre_padded:
                    dx = dx[:, :ctx.K0]
            grads[2 * i], grads[2 * i + 1] = dW, db
        return (dx, None, dy if ctx.has_add else None, None, None, None) + tuple(grads)


class FxLinear(nn.Linear):
    """nn.Linear whose forward/backward run on the fp32 MFMA GEMM (same parameters/keys)."""

    def forward(self, x, out_add=None):
        """out_add (native extension): a [rows, out_features] tensor added in the GEMM's epilogue."""
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1])
        fuse = out_add is not None and out_add.is_contiguous() \
            and out_add.shape == (x2.shape[0], self.out_features)
        head = _head_offer(self, x2.shape[0], self.out_features) \
            if (_HEAD_CTX is not None and (out_add is None or fuse)) else None
        if head is not None and _RELU_NOTES:
            note = _RELU_NOTES.pop(x2.data_ptr(), None)
            if note is not None and x2.shape[1] > note.col and x2.is_contiguous():
                head = (head[0], head[1], note)
        y = _MLPFn.apply(x2, (False,), out_add if fuse else None, None, None, head, self.weight, self.bias)
        y = y.reshape(*lead, self.out_features)
        return y + out_add if (out_add is not None and not fuse) else y


def _global_rows(count, n_local, dist):
    """Rows of the global batch behind an all-reduced statistic.  Eagerly the all-reduced count is read
    back (ranks may hold batches of different size, e.g. the last one of an epoch); while a hipGraph is
    being captured there is no host round trip and no ragged batch either (the captured step has static
    shapes on every rank): world x the local count."""
    if n_local and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        return n_local * dist.world
    return int(round(float(count.item())))


class _DiceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, alpha, mod):
        z = z.contiguous()
        N, H = z.shape
        stats = torch.empty(2 * H, dtype=torch.float32, device=z.device)
        y = torch.empty_like(z)
        ws = torch.empty(ops.dice_workspace_floats(H), dtype=torch.float32, device=z.device)
        dist = _DIST if (mod.training and _DIST is not None and _DIST.world > 1) else None
        n_total = N
        if dist is not None:
            # row-sharded training: this rank holds a slice of the global batch, the reference
            # normalises with the statistics of the WHOLE batch (activations.py:40-51) — local
            # column sums, one small all-reduce ([2H + 1] floats), then the gate
            sums = torch.empty(2 * H + 1, dtype=torch.float32, device=z.device)
            ops.dice_local_sums(z, sums, ws)
            sums[2 * H] = float(N)
            dist.all_reduce_sum(sums)
            n_total = _global_rows(sums[2 * H], N, dist)
            ops.dice_fwd_from_sums(z, alpha, mod.bn.eps, mod.bn.momentum, sums, n_total,
                                   mod.bn.running_mean, mod.bn.running_var, stats, y)
        else:
            ops.dice_fwd(z, alpha, mod.bn.eps, mod.bn.momentum, mod.training, mod.bn.running_mean,
                         mod.bn.running_var, stats, y, ws)
        ctx.save_for_backward(z, alpha, stats)
        ctx.training, ctx.eps = mod.training, mod.bn.eps
        ctx.dist, ctx.n_total = dist, n_total
        return y

    @staticmethod
    def backward(ctx, dy):
        z, alpha, stats = ctx.saved_tensors
        N, H = z.shape
        dz = torch.empty_like(z)
        ws = torch.empty(ops.dice_workspace_floats(H), dtype=torch.float32, device=z.device)
        if ctx.dist is not None:
            dy = dy.contiguous()
            sums3 = torch.empty(3 * H, dtype=torch.float32, device=z.device)
            ops.dice_bwd_local_sums(z, dy, alpha, ctx.eps, stats, sums3, ws)
            dalpha = sums3[:H].clone()          # local part; summed with the other dense gradients
            ctx.dist.all_reduce_sum(sums3[H:])  # sum dzhat, sum dzhat*zhat over the global batch
            ops.dice_bwd_from_sums(z, dy, alpha, ctx.eps, stats, sums3, ctx.n_total, dz)
            return dz, dalpha, None
        dalpha = torch.empty(H, dtype=torch.float32, device=z.device)
        ops.dice_bwd(z, dy.contiguous(), alpha, ctx.eps, ctx.training, stats, dz, dalpha, ws)
        return dz, dalpha, None


class Dice(nn.Module):
    """fuxictr/pytorch/layers/activations.py:24-51 — same submodule / parameter names
    (`bn.running_mean`, `bn.running_var`, `bn.num_batches_tracked`, `alpha`); the BatchNorm module
    only holds the running statistics, the arithmetic is fx_dice_fwd / fx_dice_bwd."""

    def __init__(self, input_dim, eps=1e-9):
        super(Dice, self).__init__()
        dev = _alloc_device()
        self.bn = nn.BatchNorm1d(input_dim, affine=False, eps=eps, momentum=0.01, device=dev)
        self.alpha = nn.Parameter(torch.zeros(input_dim, device=dev))

    def forward(self, X):
        if X.dim() != 2:
            lead = X.shape[:-1]
            return self.forward(X.reshape(-1, X.shape[-1])).reshape(*lead, X.shape[-1])
        if self.training:
            self.bn.num_batches_tracked += 1
        return _DiceFn.apply(X, self.alpha, self)


def get_activation(activation, hidden_units=None):
    """fuxictr/pytorch/torch_utils.py:137-173."""
    if isinstance(activation, str):
        if activation.lower() in ["prelu", "dice"]:
            assert type(hidden_units) == int
        if activation.lower() == "relu":
            return nn.ReLU()
        elif activation.lower() == "sigmoid":
            return nn.Sigmoid()
        elif activation.lower() == "tanh":
            return nn.Tanh()
        elif activation.lower() == "softmax":
            return nn.Softmax(dim=-1)
        elif activation.lower() == "prelu":
            return nn.PReLU(hidden_units, init=0.1)
        elif activation.lower() == "dice":
            return Dice(hidden_units)
        else:
            return getattr(nn, activation)()
    elif isinstance(activation, list):
        if hidden_units is not None:
            assert len(activation) == len(hidden_units)
            return [get_activation(act, units) for act, units in zip(activation, hidden_units)]
        return [get_activation(act) for act in activation]
    return activation


class MLP_Block(nn.Module):
    """mlp_block.py:24-96 — same `self.mlp` nn.Sequential (so the same state_dict keys); when the
    stack is only Linear / ReLU (the BASELINE configs) forward is a single fused node."""

    def __init__(self,
                 input_dim,
                 hidden_units=[],
                 hidden_activations="ReLU",
                 output_dim=None,
                 output_activation=None,
                 dropout_rates=0.0,
                 batch_norm=False,
                 bn_only_once=False,
                 use_bias=True):
        super(MLP_Block, self).__init__()
        dev = _alloc_device()
        widths = [input_dim] + list(hidden_units)
        n_hidden = len(hidden_units)
        drop = list(dropout_rates) if isinstance(dropout_rates, list) else [dropout_rates] * n_hidden
        names = hidden_activations if isinstance(hidden_activations, list) \
            else [hidden_activations] * n_hidden
        acts = get_activation(names, hidden_units)
        bn_each = batch_norm and not bn_only_once
        # module ORDER defines the state_dict keys (`mlp.<i>.weight`): [BN once] then per hidden layer
        # Linear, [BN], [activation], [Dropout]; then the optional output Linear / activation
        stack = [nn.BatchNorm1d(input_dim, device=dev)] if (batch_norm and bn_only_once) else []
        for k, (fan_in, fan_out) in enumerate(zip(widths[:-1], widths[1:])):
            stack.append(FxLinear(fan_in, fan_out, bias=use_bias, device=dev))
            if bn_each:
                stack.append(nn.BatchNorm1d(fan_out, device=dev))
            if acts[k]:
                stack.append(acts[k])
            if drop[k] > 0:
                stack.append(nn.Dropout(p=drop[k]))
        if output_dim is not None:
            stack.append(FxLinear(widths[-1], output_dim, bias=use_bias, device=dev))
        if output_activation is not None:
            stack.append(get_activation(output_activation))
        self.mlp = nn.Sequential(*stack)
        self._fused = self._fusable()

    def _fusable(self):
        """(prefix of (FxLinear, relu?) pairs, remaining modules): the Linear[/ReLU] prefix of the
        stack runs as one fused node, whatever follows (Dice, Sigmoid, ...) module by module."""
        mods = list(self.mlp)
        stack = []
        i = 0
        while i < len(mods) and isinstance(mods[i], FxLinear):
            relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
            stack.append((mods[i], relu))
            i += 2 if relu else 1
        if not stack:
            return None
        return stack, mods[i:]

    def forward(self, inputs, out_add=None, dx_into=None, out_into=None):
        """out_add (native extension, not in the reference's signature): a tensor the caller would add
        to the result anyway; when the whole stack is the fused Linear / ReLU node it rides in the last
        GEMM's epilogue.  dx_into / out_into: see _MLPFn (ignored on the unfused path; out_into only
        when the fused stack is the whole block)."""
        handed = self.__dict__.pop("_fx_dx_into", None)
        if handed is not None and handed[0] is inputs and dx_into is None:
            dx_into = handed[1]           # the record-shaped gradient buffer of the in-record attention
        pending = self.__dict__.pop("_fx_pending", None)
        if pending is not None and pending[0] is inputs and out_add is None and dx_into is None \
                and out_into is None:
            return pending[1]             # computed together with the linked CrossNetV2 (see there)
        if self._fused is None or inputs.dim() != 2:
            out = self.mlp(inputs)
            return out if out_add is None else out + out_add
        stack, tail = self._fused
        acts = tuple(r for _, r in stack)
        wb = []
        for lin, _ in stack:
            wb += [lin.weight, lin.bias]
        fuse_add = out_add is not None and not tail and not acts[-1] and out_add.is_contiguous() \
            and out_add.shape == (inputs.shape[0], stack[-1][0].weight.shape[0])
        head = None
        if _HEAD_CTX is not None and (out_add is None or fuse_add):
            # (the reference's DIN keeps the output activation INSIDE the tower, DIN.py:77-84: a deferring
            # FxSigmoid as the whole tail hands the logit on untouched)
            if not tail or (len(tail) == 1 and type(tail[0]).__name__ == "FxSigmoid"
                            and getattr(tail[0], "defer", False) and tail[0].training):
                head = _head_offer(self, inputs.shape[0], stack[-1][0].weight.shape[0])
        out = _MLPFn.apply(inputs, acts, out_add if fuse_add else None, dx_into,
                           out_into if not tail else None, head, *wb)
        for mod in tail:
            out = mod(out)
        if out_add is not None and not fuse_add:
            out = out + out_add
        return out


class _DinConcatFn(torch.autograd.Function):
    """[q, k, q-k, q*k] for every (sample, position): target_attention.py:80-82."""

    @staticmethod
    def forward(ctx, q, K):
        q = q.contiguous()
        B, L, E = K.shape
        x = torch.empty(B * L, 4 * E, dtype=torch.float32, device=q.device)
        ops.din_concat_fwd(q, K, x)
        ctx.save_for_backward(q, K)
        return x

    @staticmethod
    def backward(ctx, dx):
        q, K = ctx.saved_tensors
        B, L, E = K.shape
        dq = torch.empty(B, E, dtype=torch.float32, device=q.device)
        dK = torch.empty(B, L, E, dtype=torch.float32, device=q.device)
        ops.din_concat_bwd(dx.contiguous(), q, K, dq, dK)
        return dq, dK


class _DinPoolFn(torch.autograd.Function):
    """out[b] = sum_l w[b,l] mask[b,l] k[b,l]: target_attention.py:85-91 without softmax."""

    @staticmethod
    def forward(ctx, w, K, mask_i32):
        w = w.contiguous()
        B, L, E = K.shape
        out = torch.empty(B, E, dtype=torch.float32, device=w.device)
        ops.din_pool_fwd(w, mask_i32, K, out)
        ctx.save_for_backward(w, K, mask_i32)
        return out

    @staticmethod
    def backward(ctx, dout):
        w, K, mask_i32 = ctx.saved_tensors
        B, L, E = K.shape
        dw = torch.empty(B, L, dtype=torch.float32, device=w.device)
        dK = torch.empty(B, L, E, dtype=torch.float32, device=w.device)
        ops.din_pool_bwd(w, mask_i32, K, dout.contiguous(), dw, dK)
        return dw, dK, None


def _din_attn_forward(q, K, mask_i32, W1, b1, alpha, W2, b2, mod, out):
    """The fused attention forward (statistics pass, Dice statistics, apply pass) writing the pooled
    vector into `out` ([B, E], any row stride).  -> (stats, a, training, dist, n_total)."""
    B, L, E = K.shape
    H = W1.shape[0]
    dev = q.device
    training = mod.training
    dist = _DIST if (training and _DIST is not None and _DIST.world > 1) else None
    ws = _Workspace.get(dev, ops.din_attn_workspace_floats(B, L, E, H), tag="din_attn")
    stats = torch.empty(2 * H, dtype=torch.float32, device=dev)
    n_total = B * L
    if training:
        sums = torch.empty(2 * H + 1, dtype=torch.float32, device=dev)
        if dist is None:
            # one rank: partial sums -> statistics (+ running statistics, BatchNorm1d's step counter) in
            # the launch that finishes the sums
            ops.din_attn_stats(q, K, W1, b1, sums, ws, stats, mod.bn.momentum, mod.bn.running_mean,
                               mod.bn.running_var, mod.bn.num_batches_tracked)
        else:
            ops.din_attn_stats(q, K, W1, b1, sums, ws)
            # row-sharded training: the reference normalises with the statistics of the WHOLE
            # batch (activations.py:40-51) — one small all-reduce ([2H + 1] floats)
            sums[2 * H] = float(B * L)
            dist.all_reduce_sum(sums)
            n_total = _global_rows(sums[2 * H], B * L, dist)
            ops.dice_stats_from_sums(sums, H, n_total, mod.bn.momentum, True, mod.bn.running_mean,
                                     mod.bn.running_var, stats, mod.bn.num_batches_tracked)
    else:
        ops.dice_stats_from_sums(None, H, 1, 0.0, False, mod.bn.running_mean,
                                 mod.bn.running_var, stats)
    a = torch.empty(B, L, dtype=torch.float32, device=dev)
    ops.din_attn_fwd(q, K, W1, b1, alpha, mod.bn.eps, stats, W2.reshape(-1), b2, mask_i32, a, out)
    return stats, a, training, dist, n_total


def _din_attn_backward(q, K, mask_i32, W1, b1, alpha, W2, stats, a, eps, training, dist, n_total,
                       dout, dq, dK, dq_accumulate=False):
    """dout [B, E], dq [B, E], dK [B, L, E]: any row strides (slots of one record-shaped buffer).
    -> (dW1, db1, dalpha, dW2, db2)."""
    B, L, E = K.shape
    H = W1.shape[0]
    dev = q.device
    ws = _Workspace.get(dev, ops.din_attn_workspace_floats(B, L, E, H), tag="din_attn")
    w2 = W2.reshape(-1)
    da = torch.empty(B, L, dtype=torch.float32, device=dev)
    sums5 = torch.empty(5 * H, dtype=torch.float32, device=dev)
    ops.din_attn_bwd_sums(q, K, W1, b1, alpha, eps, stats, w2, mask_i32, dout, da, sums5, ws)
    if dist is not None:
        dist.all_reduce_sum(sums5[H:3 * H])   # sum dzhat, sum dzhat*zhat: global batch
    dW1b1 = torch.empty(H * 4 * E + H, dtype=torch.float32, device=dev)
    ops.din_attn_bwd(q, K, W1, b1, alpha, eps, training, stats, w2, mask_i32, a, dout, da, sums5,
                     n_total, dq, dK, dW1b1, ws, dq_accumulate=dq_accumulate)
    dW1 = dW1b1[:H * 4 * E].view(H, 4 * E)
    db1 = dW1b1[H * 4 * E:] if b1 is not None else None
    return dW1, db1, sums5[:H], sums5[3 * H:4 * H].view(W2.shape), sums5[4 * H:4 * H + 1]


class _DinAttnFn(torch.autograd.Function):
    """target_attention.py:66-92 with its MLP_Block(4E -> H, Dice, -> 1) as ONE autograd node on the
    fused kernels of fx_din_attn.hip: neither the [B*L, 4E] concatenation nor the [B*L, H] hidden
    tensor is written.  Forward = statistics pass, (all-reduce across ranks), Dice statistics, apply
    pass -> logits a[B, L] and the masked pooled output; backward mirrors it (sums pass -> da,
    dalpha, dW2, db2; (all-reduce); apply pass -> dq, dK, dW1, db1)."""

    @staticmethod
    def forward(ctx, q, K, mask_i32, W1, b1, alpha, W2, b2, mod):
        if q.stride(-1) != 1:
            q = q.contiguous()       # a row-strided view (a slot of the gather record) is read in place
        B, L, E = K.shape
        out = torch.empty(B, E, dtype=torch.float32, device=q.device)
        stats, a, training, dist, n_total = _din_attn_forward(q, K, mask_i32, W1, b1, alpha, W2, b2,
                                                              mod, out)
        ctx.save_for_backward(q, K, mask_i32, W1, b1, alpha, W2, stats, a)
        ctx.training, ctx.eps, ctx.dist, ctx.n_total = training, mod.bn.eps, dist, n_total
        return out

    @staticmethod
    def backward(ctx, dout):
        q, K, mask_i32, W1, b1, alpha, W2, stats, a = ctx.saved_tensors
        B, L, E = K.shape
        if dout.stride(-1) != 1:
            dout = dout.contiguous()     # (a column slice of the tower's input gradient is read in place)
        dq = torch.empty(B, E, dtype=torch.float32, device=q.device)
        dK = torch.empty(B, L, E, dtype=torch.float32, device=q.device)
        dW1, db1, dalpha, dW2, db2 = _din_attn_backward(
            q, K, mask_i32, W1, b1, alpha, W2, stats, a, ctx.eps, ctx.training, ctx.dist, ctx.n_total,
            dout, dq, dK)
        return dq, dK, None, dW1, db1, dalpha, dW2, db2, None


class _RecordGradSlot(object):
    """Where the gradient of a record prefix goes: handed to the tower as `dx_into`, it allocates the
    record-shaped gradient buffer when the tower's backward asks for its input gradient and gives out
    the prefix; the node that owns the record (_DinRecordFn) picks the buffer up again."""

    def __init__(self, B, n_slots, D, n_head, device):
        self.shape, self.n_head, self.device, self.buf = (B, n_slots, D), n_head, device, None

    def __call__(self):
        B, n_slots, D = self.shape
        self.buf = torch.empty(B, n_slots, D, dtype=torch.float32, device=self.device)
        return self.buf.view(B, n_slots * D)[:, :self.n_head * D]


class _DinRecordFn(torch.autograd.Function):
    """DIN's attention working IN the gather record (native fast path of zoo.DIN for the reference's
    configuration: one target / one sequence, the sequence the last feature).  The record is laid out
    [field 0 .. field n-1 | reserved | position 0 .. L-1] (FeatureEmbeddingDict.reserve_pooled_slot):
      forward : q = the target's slot, K = the positions, the attended vector is written into the
                reserved slot -> the tower's input [fields.., pooled] IS the record's prefix (returned
                as a view: no concatenation);
      backward: the tower writes its input gradient into the prefix of ONE record-shaped buffer
                (dx_into), the attention reads dout from the reserved slot, writes dK into the position
                slots and ADDS dq into the target's slot -> the buffer is the record's gradient as the
                embedding backward wants it: no slicing, adding or concatenating launches."""

    @staticmethod
    def forward(ctx, rec, mask_i32, W1, b1, alpha, W2, b2, mod, tslot, hole, grad_slot):
        B, n_slots, D = rec.shape
        L = n_slots - hole - 1
        q, K, out = rec[:, tslot, :], rec[:, hole + 1:, :], rec[:, hole, :]
        stats, a, training, dist, n_total = _din_attn_forward(q, K, mask_i32, W1, b1, alpha, W2, b2,
                                                              mod, out)
        ctx.save_for_backward(rec, mask_i32, W1, b1, alpha, W2, stats, a)
        ctx.training, ctx.eps, ctx.dist, ctx.n_total = training, mod.bn.eps, dist, n_total
        ctx.tslot, ctx.hole, ctx.grad_slot = tslot, hole, grad_slot
        return rec.view(B, n_slots * D)[:, :(hole + 1) * D]

    @staticmethod
    def backward(ctx, dflat):
        rec, mask_i32, W1, b1, alpha, W2, stats, a = ctx.saved_tensors
        B, n_slots, D = rec.shape
        tslot, hole, gs = ctx.tslot, ctx.hole, ctx.grad_slot
        drec, gs.buf = gs.buf, None
        if drec is None or dflat.data_ptr() != drec.data_ptr() or dflat.stride(0) != n_slots * D:
            # the tower did not write in place (unfused tower, padded input): one strided copy
            drec = torch.empty(B, n_slots, D, dtype=torch.float32, device=rec.device)
            drec.view(B, n_slots * D)[:, :(hole + 1) * D].copy_(dflat)
        q, K = rec[:, tslot, :], rec[:, hole + 1:, :]
        dW1, db1, dalpha, dW2, db2 = _din_attn_backward(
            q, K, mask_i32, W1, b1, alpha, W2, stats, a, ctx.eps, ctx.training, ctx.dist, ctx.n_total,
            drec[:, hole, :], drec[:, tslot, :], drec[:, hole + 1:, :], dq_accumulate=True)
        return drec, None, dW1, db1, dalpha, dW2, db2, None, None, None, None


def din_record_layout(emb, target, seq):
    """(rec, tslot, hole) when the embedding dict `emb` is ONE gather record laid out
    [single-slot fields.. | reserved | the positions of sequence `seq`]; None otherwise."""
    records = getattr(emb, "_records", None)
    if not records or len(records) != 1 or emb._encoded:
        return None
    rec, plan = records[0]
    hole = plan.hole.get(seq)
    if hole is None or target not in plan.slot or plan.slot[target][1] != 1 \
            or list(emb)[-1] != seq or len(emb) != hole + 1 \
            or plan.slot[seq][0] + plan.slot[seq][1] != plan.n_slots:
        return None
    for i, f in enumerate(emb):
        if f != seq and plan.slot.get(f) != (i, 1):
            return None
    return rec, plan.slot[target][0], hole


class DIN_Attention(nn.Module):
    """fuxictr/pytorch/layers/attentions/target_attention.py:26-92."""

    def __init__(self, embedding_dim=64, attention_units=[32], hidden_activations="ReLU",
                 output_activation=None, dropout_rate=0, batch_norm=False, use_softmax=False):
        super(DIN_Attention, self).__init__()
        self.embedding_dim = embedding_dim
        self.use_softmax = use_softmax
        if isinstance(hidden_activations, str) and hidden_activations.lower() == "dice":
            hidden_activations = [Dice(units) for units in attention_units]
        self.attention_layer = MLP_Block(input_dim=4 * embedding_dim, output_dim=1,
                                         hidden_units=attention_units,
                                         hidden_activations=hidden_activations,
                                         output_activation=output_activation,
                                         dropout_rates=dropout_rate, batch_norm=batch_norm)

    def _fused_plan(self):
        """(Linear 4E -> H, Dice, Linear H -> 1) when the attention MLP is exactly that — the
        reference's DIN configs — and fits fx_din_attn.hip; None otherwise (unfused kernels)."""
        plan = getattr(self, "_fx_plan", False)
        if plan is False:
            plan = None
            mods = list(self.attention_layer.mlp)
            if (_os.environ.get("FX_DIN_FUSED", "1") != "0" and not self.use_softmax
                    and len(mods) == 3 and isinstance(mods[0], FxLinear)
                    and isinstance(mods[1], Dice) and isinstance(mods[2], FxLinear)
                    and mods[2].out_features == 1
                    and mods[0].out_features <= ops.DIN_ATTN_MAX_H
                    and mods[0].in_features <= 4 * ops.DIN_ATTN_MAX_E):
                plan = (mods[0], mods[1], mods[2])
            self._fx_plan = plan
        return plan

    def _try_in_record(self, inrec, target_item, history_sequence):
        """The reference's own DIN.forward (model_zoo/DIN/src/DIN.py:118-133) reaching the in-record
        path: it calls this layer with the dict entries of the target and the sequence, replaces the
        sequence's entry by what comes back, flattens the dict and feeds the tower.  When target and
        history ARE the views of one gather record laid out [fields | reserved | positions] (link_fusion
        reserved the slot), the attention runs inside the record; the attended vector is returned as the
        view of its slot, dict2tensor hands the record's prefix to the tower and the tower writes its
        input gradient into the record-shaped gradient buffer.  -> pooled [B, D] or None (generic path)."""
        target, seq, emb_layer, dnn = inrec
        last = emb_layer.__dict__.get("_fx_last")
        if last is None:
            return None
        X, emb = last
        if emb.get(target) is not target_item or emb.get(seq) is not history_sequence:
            return None
        layout = din_record_layout(emb, target, seq)
        if layout is None:
            return None
        rec, tslot, hole = layout
        B, n_slots, D = rec.shape
        slot = _RecordGradSlot(B, n_slots, D, hole + 1, rec.device) if rec.requires_grad else None
        flat = self.forward_in_record(rec, emb_layer.packed_ids(X, seq), tslot, hole, slot)
        if flat is None:
            return None
        emb._fx_flat = (flat, seq, hole, D)
        if slot is not None:
            dnn.__dict__["_fx_dx_into"] = (flat, slot)
        return flat[:, hole * D:(hole + 1) * D]

    def forward_in_record(self, rec, mask_i32, tslot, hole, grad_slot):
        """Native fast path (see _DinRecordFn): rec [B, n_slots, D] = [fields | reserved | positions];
        -> the tower's input [B, (hole + 1) * D] as a view of the record, or None when this
        attention is not the fused configuration."""
        plan = self._fused_plan()
        if plan is None or 4 * rec.shape[2] != plan[0].in_features or mask_i32 is None \
                or mask_i32.dtype != torch.int32 or mask_i32.stride(-1) != 1:
            return None
        lin1, dice, lin2 = plan
        return _DinRecordFn.apply(rec, mask_i32, lin1.weight, lin1.bias, dice.alpha, lin2.weight,
                                  lin2.bias, dice, tslot, hole, grad_slot)

    def forward(self, target_item, history_sequence, mask=None):
        inrec = self.__dict__.get("_fx_inrec")
        if inrec is not None:
            out = self._try_in_record(inrec, target_item, history_sequence)
            if out is not None:
                return out
        seq_len = history_sequence.size(1)
        plan = self._fused_plan()
        if plan is not None and 4 * history_sequence.size(2) == plan[0].in_features \
                and target_item.dim() == 2:
            lin1, dice, lin2 = plan
            if mask is None or (mask.dtype == torch.int32 and mask.stride(-1) == 1):
                m = mask                     # e.g. the packed id columns: kept when != 0
            else:
                m = mask.to(torch.int32).contiguous()
            return _DinAttnFn.apply(target_item, history_sequence, m, lin1.weight, lin1.bias,
                                    dice.alpha, lin2.weight, lin2.bias, dice)
        if mask is not None and mask.dtype != torch.bool:
            mask = mask != 0
        attention_input = _DinConcatFn.apply(target_item, history_sequence)   # [B*L, 4E]
        attention_weight = self.attention_layer(attention_input).view(-1, seq_len)
        if self.use_softmax or history_sequence.size(2) > 64:
            # softmax variant: the reference's torch formulas (not in the BASELINE configs)
            if mask is not None:
                attention_weight = attention_weight * mask.float()
            if self.use_softmax:
                if mask is not None:
                    attention_weight = attention_weight + -1.e9 * (1 - mask.float())
                attention_weight = attention_weight.softmax(dim=-1)
            return (attention_weight.unsqueeze(-1) * history_sequence).sum(dim=1)
        if mask is None:
            m = torch.ones(attention_weight.shape, dtype=torch.int32,
                           device=attention_weight.device)
        else:
            m = mask.to(torch.int32).contiguous()
        return _DinPoolFn.apply(attention_weight, history_sequence, m)


class _CINFn(torch.autograd.Function):
    """Whole CIN stack as one autograd node (compressed_interaction_net.py:54-76); returns the
    concatenated pooled outputs [B, sum(O_i)].  args = (X0, W1, b1, W2, b2, ...), W_i = Conv1d
    weights [O_i, F0*M_i, 1]."""

    @staticmethod
    def forward(ctx, x0, *wb):
        x0 = x0.contiguous()
        B, F0, D = x0.shape
        n = len(wb) // 2
        total = sum(wb[2 * i].shape[0] for i in range(n))
        pooled = torch.empty(B, total, dtype=torch.float32, device=x0.device)
        # matrix-core shapes: every layer's W laid out once as its kernels' LDS images (one launch per 4)
        imgs, todo, Mi = [None] * n, [], F0
        for i in range(n):
            O = wb[2 * i].shape[0]
            nimg = ops.cin_wimg_floats(F0, Mi, D, O)
            if nimg:
                imgs[i] = torch.empty(nimg, dtype=torch.float32, device=x0.device)
                todo.append((wb[2 * i].view(O, -1), F0, Mi, imgs[i]))
            Mi = O
        for k in range(0, len(todo), 4):
            ops.cin_pack_w(todo[k:k + 4], D)
        xs = [x0]
        xi, off = x0, 0
        for i in range(n):
            W, b = wb[2 * i], wb[2 * i + 1]
            O = W.shape[0]
            xn = torch.empty(B, O, D, dtype=torch.float32, device=x0.device)
            ops.cin_fwd(x0, xi, W.view(O, -1), b, xn, pooled[:, off:off + O], imgs[i])
            xs.append(xn)
            xi = xn
            off += O
        ctx.wb, ctx.xs, ctx.imgs = wb, xs, imgs
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        wb, xs = ctx.wb, ctx.xs
        x0 = xs[0]
        B, F0, D = x0.shape
        n = len(wb) // 2
        dpooled = dpooled.contiguous()
        G = ops.cin_workgroups()
        dx0 = torch.empty_like(x0)
        offs, cols = [0], [0]
        for i in range(n):
            W = wb[2 * i]
            offs.append(offs[-1] + W.shape[0])
            cols.append(cols[-1] + W.shape[0] * W.shape[1] + W.shape[0])
        # the layers' per-workgroup dW / dbias sums are column slices of one buffer: one column sum at
        # the end finishes all of them
        partial = torch.empty(G, cols[-1], dtype=torch.float32, device=x0.device)
        dxn = None
        for i in range(n - 1, -1, -1):
            W = wb[2 * i]
            O = W.shape[0]
            xi = xs[i]
            dxi = torch.empty_like(xi)
            ops.cin_bwd(x0, xi, W.view(O, -1), dxn, dpooled[:, offs[i]:offs[i + 1]], dx0,
                        accumulate_dx0=(i != n - 1), dXi=dxi, partial=partial[:, cols[i]:cols[i + 1]],
                        w_img=ctx.imgs[i])
            dxn = dxi
        red = torch.empty(cols[-1], dtype=torch.float32, device=x0.device)
        ws = _Workspace.get(x0.device, _lib.FX_COLSUM_CHUNKS * cols[-1])
        ops.colsum(partial, red, ws)
        grads = [None] * (2 * n)
        for i in range(n):
            W = wb[2 * i]
            nw = W.shape[0] * W.shape[1]
            grads[2 * i] = red[cols[i]:cols[i] + nw].view(W.shape)
            grads[2 * i + 1] = red[cols[i] + nw:cols[i + 1]]
        dx0 = dx0 + dxn        # layer 1 reads X0 on both sides of the outer product
        return (dx0,) + tuple(grads)


class CompressedInteractionNet(nn.Module):
    """fuxictr/pytorch/layers/interactions/compressed_interaction_net.py:23-76 — same parameter
    containers (`cin_layer.layer_i` Conv1d, `fc` Linear) and keys; the arithmetic is fx_cin_*."""

    def __init__(self, num_fields, cin_hidden_units, output_dim=1):
        super(CompressedInteractionNet, self).__init__()
        dev = _alloc_device()
        self.cin_hidden_units = cin_hidden_units
        self.fc = FxLinear(sum(cin_hidden_units), output_dim, device=dev)
        self.cin_layer = nn.ModuleDict()
        for i, unit in enumerate(self.cin_hidden_units):
            in_channels = num_fields * self.cin_hidden_units[i - 1] if i > 0 else num_fields ** 2
            self.cin_layer["layer_" + str(i + 1)] = nn.Conv1d(in_channels, unit, kernel_size=1,
                                                              device=dev)

    def forward(self, feature_emb, out_add=None):
        """out_add (native extension): added to the result in the fc GEMM's epilogue (xDeepFM: the linear term)."""
        wb = []
        for i in range(len(self.cin_hidden_units)):
            conv = self.cin_layer["layer_" + str(i + 1)]
            wb += [conv.weight, conv.bias]
        pooled = _CINFn.apply(feature_emb, *wb)
        return self.fc(pooled, out_add=out_add)


class _CrossNetV2Fn(torch.autograd.Function):
    """X_{i+1} = X_i + X_0 * (W_i X_i + b_i), cross_net.py:126-129; one GEMM per layer with the
    bias / Hadamard / residual in its epilogue.  args = (x0, W0, b0, W1, b1, ...)."""

    @staticmethod
    def forward(ctx, x0, *wb):
        x0 = x0.contiguous()
        n = len(wb) // 2
        xs, zs = [x0], []
        xi = x0
        for i in range(n):
            W, b = wb[2 * i], wb[2 * i + 1]
            z = torch.empty_like(x0)
            xn = torch.empty_like(x0)
            ops.gemm(xi, W, xn, transa=False, transb=True, bias=b, zout=z, mul=x0, add=xi)
            xs.append(xn)
            zs.append(z)
            xi = xn
        ctx.wb, ctx.xs, ctx.zs = wb, xs, zs
        return xi

    @staticmethod
    def backward(ctx, dxn):
        wb, xs, zs = ctx.wb, ctx.xs, ctx.zs
        n = len(wb) // 2
        x0 = xs[0]
        if dxn.stride(-1) != 1:
            dxn = dxn.contiguous()         # (a row-strided slice of a cat's gradient is read in place)
        dx0 = torch.empty_like(x0)
        t = torch.empty_like(x0)
        grads = [None] * (2 * n)
        for i in range(n - 1, -1, -1):
            W, b = wb[2 * i], wb[2 * i + 1]
            # t = dxn * x0 (grad of W x_i + b); dx0 (+)= dxn * z_i; at the first layer x_i IS x_0,
            # so its residual gradient dxn joins dx0 and the last GEMM adds dx0 in its epilogue.
            ops.cross_bwd_prep(dxn, x0, zs[i], t, dx0, init=(i == n - 1), add_dxn=(i == 0))
            dW, db, dxn = linear_grads(t, xs[i], W, b is not None, add=(dx0 if i == 0 else dxn))
            grads[2 * i], grads[2 * i + 1] = dW, db
        return (dxn,) + tuple(grads)


def _dw_problem(dz, x, W, need_bias, kind):
    """-> (problem, dW, db) for dW[N_out, K_in] = dz^T x with the fused bias gradient.  kind: which tower of
    the batch ("cross" / "deep") — the two dW problems of one gemm_batch run in ONE grid and must not share
    their K-slab workspace even when their shapes coincide (a deep layer as wide as the record: ADVICE r4)."""
    Bsz, N_out = dz.shape
    K_in = x.shape[1]
    dW = torch.empty(N_out, K_in, dtype=torch.float32, device=dz.device)
    sk = max(_split_k_for(N_out, K_in, Bsz), 1 if _FORCE_SPLITK else min(8, Bsz // 256), 1)
    # one grow-only workspace per (tower, weight shape): launches follow each other on the stream
    # (ADVICE r3), the problems of one launch differ in `kind`
    ws = _Workspace.get(dz.device, ops.gemm_workspace_floats(N_out, K_in, sk), tag=("dw", kind, N_out, K_in))
    db = torch.empty(N_out, dtype=torch.float32, device=dz.device) if need_bias else None
    return ops.gemm_problem(dz, x, dW, transa=True, transb=False, split_k=sk, workspace=ws,
                            rowsum=db), dW, db


class _CrossDeepFn(torch.autograd.Function):
    """DCNv2 `model_structure: parallel` (model_zoo/DCNv2/src/DCNv2.py:108-132): CrossNetV2 over x0
    (cross_net.py:126-129) and the deep tower over x0 (mlp_block.py:96) are independent until the head,
    so layer i of one and layer i of the other leave as ONE grid (fx_gemm_f32_batch: the 160 tiles of a
    624-wide cross product fill the second workgroup slot of the CUs beside the deep layer's tiles —
    alone they occupy 62 % of the chip).  The two results are written side by side into one
    [B, D0 + H] buffer: the concatenation that feeds `fc` is never a launch.
    Backward pairs the layers from the top so that the tower with more layers finishes LAST and alone:
    its final dX takes the other tower's x0 gradient in its epilogue (no separate add).
    args = (x0, n_cross, acts, cross W0, b0, ..., deep W0, b0, ...)."""

    @staticmethod
    def forward(ctx, x0, n_cross, acts, *wb):
        # n_cross < 0: hand the two results out as TWO outputs (views of the one buffer) — the form the
        # reference's own DCNv2.forward consumes (cross_out, dnn_out, then torch.cat), see CrossNetV2
        ctx.two = n_cross < 0
        n_cross = abs(n_cross)
        x0 = x0.contiguous()
        B, D0 = x0.shape
        cwb, dwb = wb[:2 * n_cross], wb[2 * n_cross:]
        n_deep = len(acts)
        H = dwb[2 * (n_deep - 1)].shape[0]
        out = torch.empty(B, D0 + H, dtype=torch.float32, device=x0.device)
        xs, zs, hs = [x0], [], [x0]
        for i in range(max(n_cross, n_deep)):
            probs = []
            if i < n_cross:
                W, b = cwb[2 * i], cwb[2 * i + 1]
                z = torch.empty_like(x0)
                xn = out[:, :D0] if i == n_cross - 1 else torch.empty_like(x0)
                probs.append(ops.gemm_problem(xs[-1], W, xn, transb=True, bias=b, zout=z, mul=x0,
                                              add=xs[-1]))
                xs.append(xn)
                zs.append(z)
            if i < n_deep:
                W, b = dwb[2 * i], dwb[2 * i + 1]
                y = out[:, D0:] if i == n_deep - 1 else \
                    torch.empty(B, W.shape[0], dtype=torch.float32, device=x0.device)
                probs.append(ops.gemm_problem(hs[-1], W, y, transb=True, bias=b,
                                              act=1 if acts[i] else 0))
                hs.append(y)
            # (forward: no K-split problem in the batch -> the library launches the two products one
            # after the other, each with its own tile shape: measured 122 vs 125 us as one grid)
            ops.gemm_batch(probs)
        ctx.n_cross, ctx.acts, ctx.wb = n_cross, acts, wb
        ctx.xs, ctx.zs, ctx.hs, ctx.D0 = xs, zs, hs, D0
        ctx.relu_note = None
        if ctx.two:
            return out[:, :D0], out[:, D0:]
        if _HEAD_CTX is not None and acts[n_deep - 1]:
            # a fused head that reads this buffer masks the deep columns of the gradient itself
            ctx.relu_note = _RELU_NOTES[out.data_ptr()] = _ReluNote(D0)
        return out

    @staticmethod
    def backward(ctx, *douts):
        n_cross, acts, wb = ctx.n_cross, ctx.acts, ctx.wb
        xs, zs, hs, D0 = ctx.xs, ctx.zs, ctx.hs, ctx.D0
        cwb, dwb = wb[:2 * n_cross], wb[2 * n_cross:]
        n_deep = len(acts)
        x0 = xs[0]
        if ctx.two:
            # the two gradients a torch.cat backward hands out: row-strided slices of ONE buffer, read
            # in place (unit inner stride); an unused output arrives as None
            dxn, ddeep = douts
            if dxn is None:
                dxn = torch.zeros_like(x0)
            if ddeep is None:
                ddeep = torch.zeros_like(hs[n_deep])
            if dxn.stride(-1) != 1:
                dxn = dxn.contiguous()
            if ddeep.stride(-1) != 1:
                ddeep = ddeep.contiguous()
        else:
            dout = douts[0]
            if dout.stride(-1) != 1:
                dout = dout.contiguous()
            dxn = dout[:, :D0]                              # row-strided views, read in place
            ddeep = dout[:, D0:]
        note = ctx.relu_note
        if (note is not None and note.masked_ptr is not None and not ctx.two
                and douts[0].data_ptr() == note.masked_ptr
                and ddeep.stride(0) % 4 == 0 and ddeep.data_ptr() % 16 == 0):
            dz = ddeep                  # masked by the head that produced it; read in place (row-strided)
        elif acts[n_deep - 1]:
            dz = ops.mask_mul(ddeep, hs[n_deep], torch.empty_like(hs[n_deep]))
        else:
            dz = ddeep.contiguous()
        dx0 = torch.empty_like(x0)
        t = torch.empty_like(x0)
        grads = [None] * (2 * (n_cross + n_deep))
        ic, idp = n_cross - 1, n_deep - 1
        g_cross = g_deep = None        # the finished x0 gradient of a tower
        while ic >= 0 or idp >= 0:
            probs, post = [], []
            # the last layer of the tower that finishes LAST adds the other tower's x0 gradient
            if ic >= 0:
                W, b = cwb[2 * ic], cwb[2 * ic + 1]
                ops.cross_bwd_prep(dxn, x0, zs[ic], t, dx0, init=(ic == n_cross - 1),
                                   add_dxn=(ic == 0))
                pw, dW, db = _dw_problem(t, xs[ic], W, b is not None, "cross")
                dxi = torch.empty_like(x0)
                add = dx0 if ic == 0 else dxn
                probs += [pw, ops.gemm_problem(t, W, dxi, add=add)]
                grads[2 * ic], grads[2 * ic + 1] = dW, db
                post.append(("c", dxi))
            if idp >= 0:
                W, b = dwb[2 * idp], dwb[2 * idp + 1]
                pw, dW, db = _dw_problem(dz, hs[idp], W, b is not None, "deep")
                dh = torch.empty_like(hs[idp])
                mask = hs[idp] if (idp > 0 and acts[idp - 1]) else None
                add = g_cross if (idp == 0 and ic < 0 and g_cross is not None) else None
                probs += [pw, ops.gemm_problem(dz, W, dh, mask=mask, add=add)]
                grads[2 * (n_cross + idp)], grads[2 * (n_cross + idp) + 1] = dW, db
                post.append(("d", dh))
            if ic == 0 and idp < 0 and g_deep is not None:
                # cross finishes last and alone: its epilogue already adds dx0; the deep tower's x0
                # gradient joins through dx0 (one elementwise add on 10 MB, only in this rare shape)
                dx0.add_(g_deep)
                g_deep = None
            ops.gemm_batch(probs)
            for kind, val in post:
                if kind == "c":
                    dxn = val
                    if ic == 0:
                        g_cross = val
                else:
                    dz = val
                    if idp == 0:
                        g_deep = val
            if ic >= 0:
                ic -= 1
            if idp >= 0:
                idp -= 1
        # both finished in the same launch (equal depth), or cross last with the add above
        if g_cross is not None and g_deep is not None:
            used_add = (n_deep > n_cross)               # deep's last dX took g_cross in its epilogue
            dx = g_deep if used_add else g_cross + g_deep
        else:
            dx = g_cross if g_cross is not None else g_deep
        return (dx, None, None) + tuple(grads)


class CrossNetV2(nn.Module):
    """cross_net.py:95-129."""

    def __init__(self, input_dim, num_layers):
        super(CrossNetV2, self).__init__()
        self.num_layers = num_layers
        dev = _alloc_device()
        self.cross_layers = nn.ModuleList(FxLinear(input_dim, input_dim, device=dev)
                                          for _ in range(self.num_layers))

    def forward(self, X_0):
        wb = []
        for lin in self.cross_layers:
            wb += [lin.weight, lin.bias]
        partner = self.__dict__.get("_fx_partner")
        if partner is not None and self.num_layers >= 1 and X_0.dim() == 2 and X_0.shape[1] % 4 == 0:
            # the reference's DCNv2.forward (model_zoo/DCNv2/src/DCNv2.py:113-120, `parallel`): crossnet(x0),
            # then parallel_dnn(x0), then torch.cat of the two.  Both towers are computed HERE as one
            # node (cross layer i + deep layer i one grid, forward and backward); the deep tower's
            # result waits on the partner module until it is called with the same x0 (link_fusion)
            fz = partner._fused
            if fz is not None and not fz[1] and all(lin.weight.shape[0] % 4 == 0 for lin, _ in fz[0]):
                for lin, _ in fz[0]:
                    wb += [lin.weight, lin.bias]
                cross, deep = _CrossDeepFn.apply(X_0, -self.num_layers, tuple(r for _, r in fz[0]), *wb)
                partner.__dict__["_fx_pending"] = (X_0, deep)
                return cross
        return _CrossNetV2Fn.apply(X_0, *wb)



def link_fusion(model):
    """Called by BaseModel.compile(): tell the model's embedding layer which LogisticRegression
    and which FM-style interaction read the same batch, so that their work rides along in the
    embedding layer's launches (_EmbFMFn).  Only the unambiguous case is linked: ONE
    FeatureEmbeddingDict besides the one LogisticRegression owns."""
    # stock nn.Linear layers the reference's model code creates itself (DCNv2's `fc` head, DCNv2.py:100;
    # not one of the layer classes patch.install() re-binds) become FxLinear IN PLACE: same module object,
    # same Parameters and state_dict keys, forward / backward on the native GEMM (the 1648 -> 1 head on the
    # skinny kernels instead of three ATen launches).  Only the exact type, only floating-point fp32.
    if os.environ.get("FX_LINEAR_SWAP", "1") != "0":
        for mod in model.modules():
            if type(mod) is nn.Linear and mod.weight.dtype == torch.float32:
                mod.__class__ = FxLinear
    # the reference's DCNv2 with `model_structure: parallel` (DCNv2.py:86-100: attributes `crossnet`,
    # `parallel_dnn`): its forward calls the two towers one after the other on the same input.  (The
    # native mirror zoo.DCNv2 pairs them itself and keeps the concatenation out of the step as well.)
    cross, deep = getattr(model, "crossnet", None), getattr(model, "parallel_dnn", None)
    if (isinstance(cross, CrossNetV2) and isinstance(deep, MLP_Block)
            and getattr(model, "model_structure", None) == "parallel"
            and not hasattr(type(model), "_fused_parallel")
            and os.environ.get("FX_DCN_HANDOFF", "1") != "0"):
        cross.__dict__["_fx_partner"] = deep
    # the reference's DIN (DIN.py:50-100: `attention_layers`, `din_target_field`, `din_sequence_field`,
    # `embedding_layer` a FeatureEmbeddingDict, `dnn`) in its shipped configuration — one target field,
    # one raw sequence that is the LAST feature: the attention runs inside the gather record, exactly as
    # in the mirror zoo.DIN (which wires it itself: `_in_record`)
    att = getattr(model, "attention_layers", None)
    tf, sf = getattr(model, "din_target_field", None), getattr(model, "din_sequence_field", None)
    emb_layer, dnn = getattr(model, "embedding_layer", None), getattr(model, "dnn", None)
    if (att is not None and len(att) == 1 and isinstance(att[0], DIN_Attention)
            and isinstance(emb_layer, FeatureEmbeddingDict) and isinstance(dnn, MLP_Block)
            and isinstance(tf, list) and isinstance(sf, list) and len(tf) == 1 and len(sf) == 1
            and isinstance(tf[0], str) and isinstance(sf[0], str)
            and not hasattr(model, "_in_record") and os.environ.get("FX_DIN_INPLACE", "1") != "0"):
        fmap = model.feature_map.features
        names = list(fmap)
        if (names and names[-1] == sf[0] and fmap[sf[0]]["type"] == "sequence"
                and not fmap[sf[0]].get("feature_encoder")):
            emb_layer.reserve_pooled_slot(sf[0])
            att[0].__dict__["_fx_inrec"] = (tf[0], sf[0], emb_layer, dnn)
    lrs = [m for m in model.modules() if isinstance(m, LogisticRegression)]
    lr_layers = {id(lr.embedding_layer.embedding_layer) for lr in lrs}
    mains = [m for m in model.modules()
             if isinstance(m, FeatureEmbeddingDict) and id(m) not in lr_layers]
    if len(mains) != 1:
        return
    main = mains[0]
    # (through __dict__: a plain attribute assignment would register the LR module as a submodule of
    # the embedding layer and duplicate its state_dict keys)
    main.__dict__["_lr_peer"] = lrs[0] if len(lrs) == 1 else None
    main.__dict__["_fuse_fm"] = any(isinstance(m, FactorizationMachine)
                        or (isinstance(m, InnerProductInteraction)
                            and m._output_type == "product_sum") for m in model.modules())
