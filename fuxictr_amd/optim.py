"""Native optimizers: global-norm clip + Adam/SGD as fused kernels, sparse rows for the tables.

Replaces, for one training step (reference lines):
    nn.utils.clip_grad_norm_(self.parameters(), max_norm)   rank_model.py:321
    self.optimizer.step()  (torch.optim.Adam / SGD over ALL parameters, torch_utils.py:76)
The reference's tables are dense-gradient nn.Embedding, so its Adam moves every row each step.
`sparse_update="exact"` (default) reproduces exactly that trajectory while touching only the rows a
batch reads: a row's missed zero-gradient steps are replayed in registers when it is next read
(fx_adam_catchup), and `flush()` replays all rows before evaluate/save/lr change.
`sparse_update="lazy"` is torch.optim.SparseAdam semantics (rows move only when touched).
Plain SGD is identical sparse or dense.

The object is a torch.optim.Optimizer so `param_groups[*]["lr"]` (BaseModel.lr_decay,
rank_model.py:221-234) and zero_grad() keep working.
"""
import os

import torch

from . import _lib, ops
from .layers import FeatureEmbeddingDict, finish_shard_backward

# FX_ROW_RECORD=0 keeps table / m / v / last_step as four packed arrays (the layout of rounds 1 - 5) for A/B runs
ROW_RECORD = os.environ.get("FX_ROW_RECORD", "1") != "0"


class _NativeOptimizer(torch.optim.Optimizer):
    kind = None
    _require_cuda = True     # tests of the host logic lift this together with emulated kernels

    def __init__(self, params, lr, model=None, sparse_update="exact", betas=(0.9, 0.999),
                 eps=1e-8, emb_reg=None):
        if sparse_update not in ("exact", "lazy"):
            raise ValueError("sparse_update={} is not supported.".format(sparse_update))
        params = list(params)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.sparse_update = sparse_update
        self._groups = []
        self._table_param_ids = set()
        self._numeric_param_ids = set()
        self.device = None
        if model is not None:
            for mod in model.modules():
                if isinstance(mod, FeatureEmbeddingDict):
                    self._groups += mod.table_groups()
                    self._table_param_ids.update(id(p) for p in mod.table_parameters())
                    self._numeric_param_ids.update(id(p) for p in mod.numeric_parameters())
        for p in params:
            if p.is_cuda:
                self.device = p.device
                break
        if self.device is None:
            if self._require_cuda:
                raise _lib.FxError("native optimizer needs parameters on the GPU (no CPU fallback)")
            self.device = params[0].device
        # exact-mode Adam sums a row's missed zero-gradient steps from the series table behind the block
        self.scal = ops.new_scalars(self.device, lr=lr, beta1=betas[0], beta2=betas[1], eps=eps,
                                    series=(self.kind == "adam" and sparse_update == "exact"
                                            and bool(self._groups)))
        self._lr_dev = float(lr)
        self._max_norm = 0.0
        self._max_norm_explicit = False   # set_max_norm() was called by the user of the optimizer
        self._begun = False      # a step is open (zero_grad() or the first training forward opened it)
        self._begin_pending = False   # ... and its device-side part has not been launched yet
        self._model = model
        self._state_dense = {}   # id(tensor) -> (m, v)
        self._sq_dense = None
        self.dist = None
        # embedding_regularizer: [(p_norm, weight)] -> (l1, l2); non-zero switches the tables to the
        # dense-regularized step (every row has a gradient every step, rank_model.py:106-112)
        self.reg_l1 = float(sum(w for p, w in (emb_reg or []) if p == 1))
        self.reg_l2 = float(sum(w for p, w in (emb_reg or []) if p == 2))
        self.dense_reg = (self.reg_l1 != 0.0 or self.reg_l2 != 0.0) and bool(self._groups)
        if self.dense_reg:
            self.scal[_lib.SC_REG_L1:_lib.SC_REG_L1 + 1].fill_(self.reg_l1)
            self.scal[_lib.SC_REG_L2:_lib.SC_REG_L2 + 1].fill_(self.reg_l2)
        for grp in self._groups:
            self._attach(grp)
            if grp.dist is not None:
                self.dist = grp.dist
        if self.dense_reg and any(g.table is not None and g.table.dtype != torch.float32
                                  for g in self._groups):
            raise NotImplementedError("embedding_regularizer with emb_dtype=bf16 is not implemented")
        if self.dist is not None:
            # data-parallel dense side: every rank starts from rank 0's tower / numeric weights
            with torch.no_grad():
                for p in params:
                    if id(p) not in self._table_param_ids:
                        self.dist.broadcast(p.data, 0)

    # -- setup --------------------------------------------------------------------------------
    def _attach(self, grp):
        grp.scal = self.scal
        grp.opt = self
        grp.opt_kind = self.kind
        # dense_reg: every row is stepped every step, so there is nothing to catch up
        grp.exact = self.kind == "adam" and self.sparse_update == "exact" and not self.dense_reg
        grp.dense_reg = self.dense_reg
        recorded = False
        if grp.exact and grp.table is not None and ROW_RECORD:
            # round 6: [p | m | v | last_step] in one record per row (fp32 tables; bf16 tables keep four arrays)
            recorded = grp.adopt_record()
        if not recorded:
            grp.drop_record()
            if self.kind == "adam" and grp.table is not None:
                grp.m = torch.zeros_like(grp.table, dtype=torch.float32)     # fp32 state, also for
                grp.v = torch.zeros_like(grp.table, dtype=torch.float32)     # bf16 tables
            if (self.kind == "adam" or self.dense_reg) and grp.table is not None:
                grp.last_step = torch.zeros(grp.table.shape[0], dtype=torch.int32, device=grp.device)
        if self.dense_reg and grp.table is not None:
            grp.reg_partials = torch.zeros(3 * _lib.FX_REG_BLOCKS, dtype=torch.float32,
                                           device=grp.device)
            grp.reg_cross = torch.zeros(_lib.FX_REG_CROSS_BLOCKS, dtype=torch.float32,
                                        device=grp.device)
            grp.reg_fresh = False

    def set_max_norm(self, max_norm, _from_model=False):
        """Global-norm clip applied inside the update kernels (0: no clipping).  An explicit call wins
        over the model's `_max_gradient_norm` (which step() only adopts while nobody has set one: a
        train_step override that deliberately does not clip calls set_max_norm(0)); set_max_norm(None)
        clears the explicit value and returns control to the model."""
        if not _from_model:
            if max_norm is None:
                # hand control back to the model: step() adopts fit()'s max_gradient_norm again
                self._max_norm_explicit = False
                if self._model is not None and hasattr(self._model, "_max_gradient_norm"):
                    max_norm = self._model._max_gradient_norm
            else:
                self._max_norm_explicit = True
        max_norm = float(max_norm) if max_norm else 0.0
        if max_norm != self._max_norm:
            self.scal[_lib.SC_MAX_NORM:_lib.SC_MAX_NORM + 1].fill_(max_norm)
            self._max_norm = max_norm

    def sync_lr(self):
        """Push a host-side lr change (lr_decay) to the device; exact mode first replays every
        pending row with the OLD lr, since the replay uses the lr in the scalar block."""
        lr = float(self.param_groups[0]["lr"])
        if lr != self._lr_dev:
            self.flush()
            self.scal[_lib.SC_LR:_lib.SC_LR + 1].fill_(lr)
            self._lr_dev = lr

    def begin_step(self):
        """Opens a training step BEFORE its forward: t += 1 and the Adam bias corrections on the
        device (exact mode replays a row's missed steps up to t - 1 when the forward reads it).
        Idempotent until step() closes the step.  Three callers, so that both loop shapes of the
        reference advance t correctly: zero_grad() (BaseModel.train_step: zero_grad; forward; backward;
        step — rank_model.py:307-323), the first training forward of a step (`ensure_begun`, for the
        LongCTR models' forward; backward; step; zero_grad order, model_zoo/LongCTR/DCNv2/DCNv2.py:209-217,
        whose first step has no zero_grad before it) and step() itself (a model without table lookups)."""
        if self._begun:
            return
        self.sync_lr()
        # the device-side part (t += 1, bias corrections) rides along in the first de-dup launch of
        # the step (fx_dedup_catchup); whoever needs it earlier calls flush_begin()
        self._begin_pending = True
        self._begun = True

    def ensure_begun(self):
        """Called by the embedding layers at the top of a training forward: opens the step if nobody
        has, and re-reads the host-side lr while the device-side opening is still pending (zero_grad()
        at the END of the previous step opened this one before an lr_decay between epochs)."""
        if not self._begun:
            self.begin_step()
        elif self._begin_pending:
            self.sync_lr()

    def take_begin(self):
        """-> the scalar block if the step's device-side opening is still due (the caller's kernel
        performs it), else None."""
        if self._begin_pending:
            self._begin_pending = False
            return self.scal
        return None

    def flush_begin(self):
        if self._begin_pending:
            self._begin_pending = False
            ops.opt_begin_step(self.scal)

    def flush(self):
        for grp in self._groups:
            grp.flush()

    def check_errors(self):
        """Host sync: raise if a kernel saw an id outside its table (reference: IndexError)."""
        flag = int(self.scal.view(torch.int32)[_lib.SC_ERR].item())
        if flag & _lib.FX_FLAG_BAD_ID:
            raise IndexError("embedding id out of range (native gather flagged FX_FLAG_BAD_ID)")
        if flag & _lib.FX_FLAG_A2A_OVERFLOW:
            raise RuntimeError("row-sharded exchange overflowed its per-peer capacity: raise "
                               "`a2a_capacity_factor` (>= world size can never overflow)")

    # -- embedding regularizer ----------------------------------------------------------------
    def _reg_grad(self, w):
        r = self.reg_l2 * w
        if self.reg_l1:
            r = r + self.reg_l1 * torch.sign(w)
        return r

    @torch.no_grad()
    def emb_reg_loss(self):
        """sum over the FeatureEmbeddingDict parameters of l1 |p|_1 + l2/2 |p|_2^2 (device scalar,
        no autograd: its gradient is applied inside the update kernels)."""
        if not self.dense_reg:
            return 0
        nb = _lib.FX_REG_BLOCKS
        sq = torch.empty(1, dtype=torch.float32, device=self.device)
        ab = torch.empty(1, dtype=torch.float32, device=self.device)
        sqs, abs_ = [], []
        for grp in self._groups:
            if grp.table is not None:
                ops.reg_stats(grp.table, self.scal, grp.reg_partials)
                grp.reg_fresh = True
                sqs.append(grp.reg_partials[:nb])
                abs_.append(grp.reg_partials[nb:2 * nb])
        ops.sum_parts(sqs, sq)
        ops.sum_parts(abs_, ab)
        if self.dist is not None:
            # row-sharded tables: every rank summed ITS rows; the loss term is the global sum
            both = torch.cat([sq, ab])
            self.dist.all_reduce_sum(both)
            sq, ab = both[0:1], both[1:2]
        term = (0.5 * self.reg_l2) * sq[0] + self.reg_l1 * ab[0]
        for grp in self._groups:
            if grp.num_w is not None:
                term = term + (0.5 * self.reg_l2) * (grp.num_w * grp.num_w).sum() \
                    + self.reg_l1 * grp.num_w.abs().sum()
        return term

    # -- step ---------------------------------------------------------------------------------
    def _dense_lists(self):
        ps, gs = [], []
        for group in self.param_groups:
            for p in group["params"]:
                if id(p) in self._table_param_ids or id(p) in self._numeric_param_ids:
                    continue
                if p.grad is None:
                    continue
                g = p.grad
                if not g.is_contiguous():
                    g = g.contiguous()
                if not p.is_contiguous():
                    raise _lib.FxError("non-contiguous dense parameter")
                ps.append(p.data)
                gs.append(g)
        for grp in self._groups:
            if grp.num_w is None:
                continue
            if self.dense_reg:
                g = self._reg_grad(grp.num_w)
                if self.dist is not None:
                    # replicated weights: every rank adds the same term and the dense gradients are
                    # SUM-reduced (the data part was pre-scaled by 1/world through the loss)
                    g = g / self.dist.world
                ps.append(grp.num_w)
                gs.append(g if grp.num_grad is None else g + grp.num_grad)
            elif grp.num_grad is not None:
                ps.append(grp.num_w)
                gs.append(grp.num_grad)
        return ps, gs

    def _moments(self, ps):
        ms, vs = [], []
        for p in ps:
            st = self._state_dense.get(p.data_ptr())
            if st is None:
                st = (torch.zeros_like(p), torch.zeros_like(p))
                self._state_dense[p.data_ptr()] = st
            ms.append(st[0])
            vs.append(st[1])
        return ms, vs

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("closure is not supported by the native optimizer")
        if not self._begun:
            # no zero_grad() / training forward opened the step (a model without table lookups, or an
            # accidental second step()): the step counter and the bias corrections advance
            if self._groups and not getattr(self, "_warned_open", False):
                import logging
                logging.warning("native optimizer: step() had to open the step itself (no zero_grad() or "
                                "training forward since the last step) - t advances once more")
                self._warned_open = True
            self.begin_step()
        self._begun = False
        self.flush_begin()
        if (not self._max_norm_explicit and self._model is not None
                and hasattr(self._model, "_max_gradient_norm")):
            # the global-norm clip (rank_model.py:321) lives inside the update kernels: a train_step
            # override that reaches step() directly still gets fit()'s max_gradient_norm, table
            # gradients included (they are invisible to a clip_grad_norm_ over .grad attributes);
            # an explicit optimizer.set_max_norm() is never overridden
            self.set_max_norm(self._model._max_gradient_norm, _from_model=True)
        if self.dist is not None:           # row-gradient all-to-all(s) + owner-side reduction
            finish_shard_backward([grp for grp in self._groups if grp.dist is not None])
        ps, gs = self._dense_lists()
        tsq = None
        if self.dist is not None:
            for grp in self._groups:
                if len(grp.pending) > 1:
                    raise NotImplementedError("several lookups of one table group per step")
            # ONE all-reduce per step: every dense gradient (losses were pre-scaled by 1/world)
            # followed by this rank's partial sums of the table part of the squared gradient norm (table
            # rows are disjoint across ranks, so the element-wise SUM of the partial arrays still adds
            # up to the global table part; no reduction launch before the collective)
            tparts = [rec.sq.reshape(-1) for grp in self._groups for rec in grp.pending
                      if rec.sq is not None]
            if self.dense_reg:
                # |G + r|^2 over this rank's rows = sum r^2 + sum G^2 + sum 2 G.r (see fx_reg_*)
                for grp in self._groups:
                    if grp.table is None:
                        continue
                    if not grp.reg_fresh:
                        ops.reg_stats(grp.table, self.scal, grp.reg_partials)
                    grp.reg_fresh = False
                    tparts.append(grp.reg_partials[2 * _lib.FX_REG_BLOCKS:])
                    for rec in grp.pending:
                        ops.reg_cross(grp.table, grp.D, rec.dd, rec.G, self.scal, grp.reg_cross)
                        tparts.append(grp.reg_cross.clone())
            # every gradient starts on a 16-byte boundary of the flat buffer (a 1-element bias in the
            # middle would otherwise push everything after it onto the scalar path of the
            # multi-tensor norm / update kernels: 36 us instead of 19 us for k_mt_adam)
            if getattr(self, "_flat_pad", None) is None:
                self._flat_pad = torch.zeros(3, dtype=torch.float32, device=self.device)
            pieces, offs, off = [], [], 0
            for g in gs:
                pieces.append(g.reshape(-1))
                offs.append(off)
                off += g.numel()
                if off % 4:
                    pieces.append(self._flat_pad[:4 - off % 4])
                    off += 4 - off % 4
            n_t = sum(t.numel() for t in tparts)
            flat = torch.cat(pieces + tparts)
            self.dist.all_reduce_sum(flat)
            gs = [flat[o:o + g.numel()].view_as(g) for o, g in zip(offs, gs)]
            tsq = flat[off:off + n_t]
        parts = []
        if ps:
            need = len(ps) * _lib.FX_MT_BLOCKS
            if self._sq_dense is None or self._sq_dense.numel() != need:
                self._sq_dense = torch.empty(need, dtype=torch.float32, device=self.device)
            ops.mt_sqnorm(gs, self._sq_dense)
            parts.append(self._sq_dense)
        for grp in self._groups:
            if len(grp.pending) > 1:
                raise NotImplementedError(
                    "a table group was looked up %d times in one training step; merging several "
                    "sparse gradients per step is not implemented" % len(grp.pending))
            if tsq is None:
                for rec in grp.pending:
                    if rec.sq is not None:
                        parts.append(rec.sq)
            if self.dense_reg and grp.table is not None and tsq is None:
                if not grp.reg_fresh:
                    ops.reg_stats(grp.table, self.scal, grp.reg_partials)
                grp.reg_fresh = False
                parts.append(grp.reg_partials[2 * _lib.FX_REG_BLOCKS:])
                for rec in grp.pending:
                    ops.reg_cross(grp.table, grp.D, rec.dd, rec.G, self.scal, grp.reg_cross)
                    parts.append(grp.reg_cross)
        if tsq is not None and tsq.numel():
            # global norm^2 = dense part (identical on every rank after the all-reduce) + the
            # all-reduced table part
            parts.append(tsq)
        ops.clip_coef(parts, self.scal)
        if ps:
            self._dense_update(ps, gs)
        # table groups that were reduced against the SAME de-dup result (the D=16 tables and the D=1
        # tables of LogisticRegression) are updated by one launch
        buckets = {}
        for grp in self._groups:
            for rec in grp.pending:
                buckets.setdefault(id(rec.dd), []).append((grp, rec))
        for items in buckets.values():
            if len(items) == 1 and items[0][0].table.dtype == torch.float32 and items[0][0].record is None:
                self._sparse_update(*items[0])
                continue
            # dtype-aware multi-table kernel, FX_MAX_TABLES groups per launch (a bf16 table must never
            # reach the fp32-only single-table kernels)
            for i in range(0, len(items), _lib.FX_MAX_TABLES):
                part = items[i:i + _lib.FX_MAX_TABLES]
                ops.sparse_update_multi(self.kind, [g.row_state(G=r.G) for g, r in part],
                                        part[0][1].dd, self.scal)
        for grp in self._groups:
            if self.dense_reg and grp.table is not None:
                ops.reg_dense_update(grp.table, grp.m, grp.v, grp.last_step, grp.D,
                                     self.kind == "adam", self.scal)
            grp.pending = []
            grp.num_grad = None
        return None

    # -- checkpoint / resume (the reference saves weights only) ------------------
    def state_dict(self):
        """Step counter, lr, dense moments (by position in param_groups) and each table group's
        row moments / last-step stamps — of THIS rank's shard when the tables are row-sharded."""
        self.flush()                      # exact mode: no pending replays in a checkpoint
        step = int(self.scal.view(torch.int32)[_lib.SC_STEP].item())
        dense = []
        for group in self.param_groups:
            for p in group["params"]:
                st = self._state_dense.get(p.data_ptr())
                dense.append(None if st is None else (st[0].cpu(), st[1].cpu()))
        groups = []
        for grp in self._groups:
            numst = None
            if grp.num_w is not None:
                st = self._state_dense.get(grp.num_w.data_ptr())
                numst = None if st is None else (st[0].cpu(), st[1].cpu())
            groups.append({"m": None if grp.m is None else grp.m.cpu(),
                           "v": None if grp.v is None else grp.v.cpu(),
                           "last_step": None if grp.last_step is None else grp.last_step.cpu(),
                           "num": numst})
        return {"fx_step": step, "fx_lr": float(self.param_groups[0]["lr"]), "fx_kind": self.kind,
                "fx_dense": dense, "fx_groups": groups}

    def load_state_dict(self, state):
        if state.get("fx_kind") != self.kind:
            raise ValueError("optimizer state of kind %r loaded into %r" % (state.get("fx_kind"),
                                                                            self.kind))
        self.scal.view(torch.int32)[_lib.SC_STEP:_lib.SC_STEP + 1].fill_(int(state["fx_step"]))
        for group in self.param_groups:
            group["lr"] = state["fx_lr"]
        self.scal[_lib.SC_LR:_lib.SC_LR + 1].fill_(state["fx_lr"])
        self._lr_dev = float(state["fx_lr"])
        it = iter(state["fx_dense"])
        for group in self.param_groups:
            for p in group["params"]:
                st = next(it)
                if st is not None:
                    self._state_dense[p.data_ptr()] = (st[0].to(p.device), st[1].to(p.device))
        for grp, gs in zip(self._groups, state["fx_groups"]):
            for name in ("m", "v", "last_step"):
                if gs[name] is not None and getattr(grp, name) is not None:
                    getattr(grp, name).copy_(gs[name])
            if gs["num"] is not None and grp.num_w is not None:
                self._state_dense[grp.num_w.data_ptr()] = (gs["num"][0].to(grp.device),
                                                           gs["num"][1].to(grp.device))

    def zero_grad(self, set_to_none=True):
        super().zero_grad(set_to_none=set_to_none)
        for grp in self._groups:
            grp.pending = []
            grp.num_grad = None
        self.begin_step()


class NativeAdam(_NativeOptimizer):
    """torch.optim.Adam(params, lr) defaults (betas 0.9/0.999, eps 1e-8, no weight decay)."""
    kind = "adam"

    def _dense_update(self, ps, gs):
        ms, vs = self._moments(ps)
        ops.mt_adam(ps, gs, ms, vs, self.scal)

    def _sparse_update(self, grp, rec):
        ops.sparse_adam(grp.table, grp.m, grp.v, grp.last_step, grp.D, rec.dd, rec.G, self.scal)


class NativeSGD(_NativeOptimizer):
    """torch.optim.SGD(params, lr) defaults (no momentum, no weight decay)."""
    kind = "sgd"

    def _dense_update(self, ps, gs):
        ops.mt_sgd(ps, gs, self.scal)

    def _sparse_update(self, grp, rec):
        ops.sparse_sgd(grp.table, grp.D, rec.dd, rec.G, self.scal, last_step=grp.last_step)


def get_optimizer(optimizer, params, lr, model=None, sparse_update="exact", emb_reg=None):
    """fuxictr/pytorch/torch_utils.py:58-79 — string -> optimizer; Adam and SGD are native."""
    if isinstance(optimizer, str):
        name = optimizer.lower()
        if name == "adam":
            return NativeAdam(params, lr, model=model, sparse_update=sparse_update, emb_reg=emb_reg)
        if name == "sgd":
            return NativeSGD(params, lr, model=model, sparse_update=sparse_update, emb_reg=emb_reg)
    raise NotImplementedError("optimizer={} is not supported.".format(optimizer))
