"""`BaseModel` — host-side mirror of fuxictr/pytorch/models/rank_model.py:31-470 with the
training step re-plumbed onto the native kernels.

Kept verbatim (names, arguments, behaviour): ctor kwargs, compile, fit, train_epoch, train_step,
evaluate, predict, lr_decay, checkpoint_and_earlystop, save/load_weights, count_parameters, and
the per-model contract forward(inputs) -> {"y_pred": Tensor[B,1]}.

What changed underneath train_step (rank_model.py:307-323):
    zero_grad / forward / loss / backward     same Python shape, native autograd nodes
    clip_grad_norm_ + optimizer.step()        fused into the native optimizer (optim.py)
    sigmoid + binary_cross_entropy            one fused kernel (fx_sigmoid_bce)
"""
import logging
import os
import sys
from collections import OrderedDict

import numpy as np
import torch
from torch import nn

from . import _lib, layers, ops
from .layers import FeatureDict, FeatureEmbeddingDict, FxLinear, _NumericView, not_in_whitelist
from .optim import get_optimizer


# -- small host utilities mirrored from fuxictr/utils.py:166-206 and fuxictr/metrics.py:26-51 ------
class Monitor(object):
    def __init__(self, kv):
        if isinstance(kv, str):
            kv = {kv: 1}
        self.kv_pairs = kv

    def get_value(self, logs):
        return sum(logs.get(k, 0) * v for k, v in self.kv_pairs.items())

    def get_metrics(self):
        return list(self.kv_pairs.keys())


def evaluate_metrics(y_true, y_pred, metrics, group_id=None):
    """logloss / AUC through scikit-learn on float64, exactly as the reference (metrics.py:49-51),
    so "AUC to 4 decimals" compares like with like.  Group metrics are out of scope."""
    from sklearn.metrics import log_loss, roc_auc_score
    out = OrderedDict()
    for metric in metrics:
        if metric in ["logloss", "binary_crossentropy"]:
            out[metric] = log_loss(y_true, y_pred)
        elif metric == "AUC":
            out[metric] = roc_auc_score(y_true, y_pred)
        elif metric in ["gAUC", "avgAUC", "MRR"] or metric.startswith("NDCG"):
            raise NotImplementedError("metrics={} not implemented.".format(metric))
        else:
            raise ValueError("metric={} not supported.".format(metric))
    return out


def get_device(gpu=-1):
    """torch_utils.py:42-56, except that the native path REQUIRES a GPU."""
    if gpu >= 0 and torch.cuda.is_available():
        return torch.device("cuda:" + str(gpu))
    raise _lib.FxError("fuxictr_amd runs on an MI355X only: pass gpu>=0 on a machine with a "
                       "visible HIP device (gpu=%s, cuda available=%s)"
                       % (gpu, torch.cuda.is_available()))


def get_regularizer(reg):
    """torch_utils.py:106-135."""
    reg_pair = []
    if isinstance(reg, float):
        reg_pair.append((2, reg))
    elif isinstance(reg, str):
        try:
            if reg.startswith("l1(") or reg.startswith("l2("):
                reg_pair.append((int(reg[1]), float(reg.rstrip(")").split("(")[-1])))
            elif reg.startswith("l1_l2"):
                l1_reg, l2_reg = reg.rstrip(")").split("(")[-1].split(",")
                reg_pair.append((1, float(l1_reg)))
                reg_pair.append((2, float(l2_reg)))
            else:
                raise NotImplementedError
        except Exception:
            raise NotImplementedError("regularizer={} is not supported.".format(reg))
    return reg_pair


# -- fused output activation + loss ----------------------------------------------------------------
class _SigmoidFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logit):
        logit = logit.contiguous()
        p = torch.empty_like(logit)
        ops.sigmoid_bce(logit, None, prob=p)
        ctx.save_for_backward(p)
        return p

    @staticmethod
    def backward(ctx, dp):
        (p,) = ctx.saved_tensors
        return dp * p * (1 - p)  # only reached by custom losses; the BCE path never comes here


class FxSigmoid(nn.Module):
    """nn.Sigmoid whose input logit is remembered so that the BCE loss can be fused with it.
    `defer` (set by BaseModel around a training step whose loss is the fused BCE): the
    probabilities are not needed by anyone, skip their kernel — `y_pred` then carries the logits
    and must only be consumed through `_fx_logit`."""
    defer = False

    def forward(self, x):
        if self.defer and self.training:
            p = x.view_as(x)
            p._fx_logit = x
            p._fx_deferred = True
            return p
        p = _SigmoidFn.apply(x)
        p._fx_logit = x
        return p


_UNIT_GRADS = {}


def _unit_grad(device):
    """Persistent 1.0 used as the root gradient of loss.backward(): saves autograd's ones_like
    fill, and lets _SigmoidBCEFn.backward recognise 'multiply by one' without reading the value."""
    g = _UNIT_GRADS.get(device)
    if g is None:
        g = _UNIT_GRADS[device] = torch.ones((), dtype=torch.float32, device=device)
    return g


_SCALED_GRADS = {}


def _scaled_grad(device, world):
    """Persistent 1/world root gradient of the sharded step (global-batch mean = mean of the ranks'
    local means): one multiply in the loss backward instead of div + ones_like + div-backward."""
    g = _SCALED_GRADS.get((device, world))
    if g is None:
        g = _SCALED_GRADS[(device, world)] = torch.full((), 1.0 / world, dtype=torch.float32,
                                                        device=device)
    return g


class _SigmoidBCEFn(torch.autograd.Function):
    """loss = mean BCE(sigmoid(logit), y); forward also produces dloss/dlogit."""

    @staticmethod
    def forward(ctx, logit, y):
        logit = logit.contiguous()
        y = y.contiguous()
        loss = torch.empty((), dtype=torch.float32, device=logit.device)
        dlogit = torch.empty_like(logit)
        ops.sigmoid_bce(logit, y, loss=loss, dlogit=dlogit)
        ctx.save_for_backward(dlogit)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dlogit,) = ctx.saved_tensors
        unit = _UNIT_GRADS.get(g.device)
        if unit is not None and g.data_ptr() == unit.data_ptr():
            return dlogit, None
        return dlogit * g, None


class _FusedHeadLossFn(torch.autograd.Function):
    """The loss of a step whose tower head already evaluated it (layers._HeadCtx, ops.head_train): forward
    hands out that value, backward the dlogit formed with it (already times the root gradient the step
    announced — any other root gradient is an error, the tower's gradients below were built on it)."""

    @staticmethod
    def forward(ctx, logit, dlogit, loss, root_ptr):
        ctx.dlogit, ctx.root_ptr = dlogit, root_ptr
        return loss.detach()

    @staticmethod
    def backward(ctx, g):
        if g.data_ptr() != ctx.root_ptr:
            raise RuntimeError("fused training head: loss.backward() got a root gradient other than the one "
                               "BaseModel._forward_backward announced (set FX_HEAD_FUSED=0 for custom loops "
                               "that scale the loss)")
        return ctx.dlogit, None, None, None


def _bce_loss(y_pred, y_true, reduction="mean"):
    logit = getattr(y_pred, "_fx_logit", None)
    if logit is not None and reduction == "mean":
        hc = layers._HEAD_CTX
        # (same storage AND same version: a logit that was modified in place after the head ran — the
        # reference's AutoInt / DESTINE `y_pred += ...` — is not the value the fused loss was formed on)
        if hc is not None and hc.result is not None and hc.result[0].data_ptr() == logit.data_ptr() \
                and hc.result[0].numel() == logit.numel() and hc.y.data_ptr() == y_true.data_ptr() \
                and logit._version == hc.result[3] and hc.result[0]._version == hc.result[3]:
            return _FusedHeadLossFn.apply(logit, hc.result[1].view(logit.shape), hc.result[2], hc.root_ptr)
        return _SigmoidBCEFn.apply(logit, y_true)
    if getattr(y_pred, "_fx_deferred", False):
        y_pred = torch.sigmoid(logit)
    return torch.nn.functional.binary_cross_entropy(y_pred, y_true, reduction=reduction)


def get_loss(loss):
    """torch_utils.py:81-104."""
    if isinstance(loss, str):
        if loss in ["bce", "binary_crossentropy", "binary_cross_entropy"]:
            return _bce_loss
        try:
            return getattr(torch.nn.functional, loss)
        except AttributeError:
            raise NotImplementedError("loss={} is not supported.".format(loss))
    return loss


class _GraphStep(object):
    """One captured training step with static input buffers."""

    def __init__(self, model, batch):
        self.model = model
        dev = model.device
        label = model.feature_map.labels[0]
        self.label = label
        self.B = batch[label].shape[0]
        # which packed id / dense matrices the model's embedding layers ask for: run one eager
        # step on a FeatureDict and read its cache
        probe = model.get_inputs(batch)
        probe._fx_ready = True
        probe[label] = batch[label].to(dev)
        self.probe_loss = model._side_stream_step(probe)
        self.packs = []      # (sig, id feature names, numeric feature names, ids, dense)
        static = FeatureDict()
        static._fx_ready = True
        for key, val in probe.cache.items():
            if key[0] != "pack":
                continue
            ids, dense = val
            id_feats, num_feats = key[1]
            s_ids = torch.zeros_like(ids) if ids is not None else None
            s_dense = torch.zeros_like(dense) if dense is not None else None
            self.packs.append((key, [f for f, _ in id_feats], list(num_feats), s_ids, s_dense))
            static.cache[key] = (s_ids, s_dense)
            col = 0
            for f, w in id_feats:
                static[f] = s_ids[:, col] if w == 1 else s_ids[:, col:col + w]
                col += w
            for j, f in enumerate(num_feats):
                static[f] = s_dense[:, j]
        for f in probe.keys():
            if f not in static and f != label:
                static[f] = probe[f].clone()
        self.y = torch.zeros(self.B, 1, dtype=torch.float32, device=dev)
        static[label] = self.y.view(-1)
        self.static = static
        self._pack_keys = {k for k, *_ in self.packs}
        self._fill_names = sorted({f for _, ids_n, num_n, _, _ in self.packs for f in ids_n + num_n})
        self._fill_cache = {}
        self._fill_stream = []       # ids of the cache entries that came in through train_step, oldest first
        self.fill(batch)
        torch.cuda.synchronize(dev)
        if model._dist is None:
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=model._graph_stream):
                self.loss = model._step_body(static)
        else:
            # row-sharded: hipGraph segments with the collectives launched eagerly in between
            import gc
            from .dist import GraphSegments
            nccl = model._dist.backend == "nccl"

            def record():
                self.graph = GraphSegments(comm_stream=torch.cuda.Stream(dev) if nccl else None,
                                           settle_s=0.25 if nccl else 0.0)
                gc.collect()
                torch.cuda.empty_cache()
                cur = torch.cuda.current_stream(dev)
                model._graph_stream.wait_stream(cur)
                with torch.cuda.stream(model._graph_stream):
                    model._dist.recorder = self.graph
                    try:
                        self.graph.begin()
                        self.loss = model._step_body(static)
                        self.graph.finish()
                    except BaseException:
                        # end the capture HERE, while the capturing stream is still the current one
                        # (capture_end on another stream raises and leaves this one capturing for good)
                        self.graph.abort()
                        raise
                    finally:
                        model._dist.recorder = None
                cur.wait_stream(model._graph_stream)
                torch.cuda.synchronize(dev)
            try:
                record()
            except Exception as exc:   # noqa: BLE001 — a stack that cannot record RCCL kernels
                if not model._dist.capture_collectives:
                    raise
                logging.warning("recording the collectives into the step's hipGraph failed (%s: %s); "
                                "hipGraph segments with eager collectives instead",
                                type(exc).__name__, exc)
                model._dist.capture_collectives = False
                # per-batch entries the aborted recording left in the static batch's cache
                static.cache = {k: v for k, v in static.cache.items() if k in self._pack_keys}
                model.optimizer._begun = False
                model.optimizer._begin_pending = False
                for grp in model.optimizer._groups:
                    grp.pending, grp.num_grad, grp._await_exchange = [], None, []
                torch.cuda.synchronize(dev)
                record()
            model._dist.graph_mode = ("one hipGraph per step with the RCCL collectives recorded in it"
                                      if model._dist.capture_collectives else
                                      "hipGraph segments with the collectives launched between them")
        # the capture only recorded the step; drop per-batch caches created while recording
        static.cache = {k: v for k, v in static.cache.items() if k in self._pack_keys}

    FILL_CACHE_MAX = 512       # batches registered through BaseModel.prepare_batch (bench.py's pool, epochs
    #                            over device-resident batches): the caller keeps those alive anyway
    FILL_CACHE_STREAM = 4      # batches first seen by train_step: a loop that streams fresh device-resident
    #                            batches must not pin hundreds of them — a short LRU

    def fill(self, batch, launch=True, registered=False):
        """Cast the batch's columns + label into the static input buffers (one launch).  The host
        side of that launch (40 tensor look-ups, the ctypes argument blocks) is kept per batch OBJECT
        whose columns are device-resident: a batch seen before (an epoch over resident batches,
        bench.py's pool, anything handed to BaseModel.prepare_batch) costs one call, as long as the
        dict still holds the very same tensors."""
        ent = self._fill_cache.get(id(batch))
        if ent is not None:
            if registered and id(batch) in self._fill_stream:
                self._fill_stream.remove(id(batch))        # promoted: prepare_batch() vouches for it
            if ent[0] is batch and all(batch.get(f) is t for f, t in ent[1]):
                self.model._staged_labels = (None, None)
                if launch:
                    ent[2]()
                return
            del self._fill_cache[id(batch)]
        dev = self.model.device
        staged = self.model._stage_host_columns(batch, self._fill_names)   # one async H2D copy per dtype
        key, y = self.model._staged_labels

        def col(f):
            return staged[f] if f in staged else batch[f].to(dev)
        items, srcs = [], []
        for _, id_names, num_names, s_ids, s_dense in self.packs:
            for dst, cols_ in ((s_ids, id_names), (s_dense, num_names)):
                c0 = 0
                for f in cols_ if dst is not None else ():
                    t = col(f)
                    items.append((t, dst, c0))
                    srcs.append((f, t))
                    c0 += 1 if t.dim() == 1 else t.shape[1]
        yt = y if key == id(batch) else batch[self.label].to(dev)
        items.append((yt, self.y, 0))
        srcs.append((self.label, yt))
        if (not staged and type(batch) is dict
                and all(batch.get(f) is t and t.is_contiguous() for f, t in srcs)):
            call = ops.pack_columns_multi_prepare(items)
            if registered:
                if len(self._fill_cache) >= self.FILL_CACHE_MAX:       # oldest entry out
                    self._fill_cache.pop(next(iter(self._fill_cache)))
            else:
                self._fill_stream.append(id(batch))
                if len(self._fill_stream) > self.FILL_CACHE_STREAM:
                    self._fill_cache.pop(self._fill_stream.pop(0), None)
            self._fill_cache[id(batch)] = (batch, srcs, call)
            if launch:
                call()
            return
        if launch:
            ops.pack_columns_multi(items)      # ids + numerics + label of the batch: one launch


class BaseModel(nn.Module):
    def __init__(self,
                 feature_map,
                 model_id="BaseModel",
                 task="binary_classification",
                 gpu=-1,
                 monitor="AUC",
                 save_best_only=True,
                 monitor_mode="max",
                 early_stop_patience=2,
                 eval_steps=None,
                 embedding_regularizer=None,
                 net_regularizer=None,
                 reduce_lr_on_plateau=True,
                 **kwargs):
        super(BaseModel, self).__init__()
        self.device = get_device(gpu)
        torch.cuda.set_device(self.device)
        layers.set_default_device(self.device)  # native layers allocate their tables here
        shard = kwargs.get("shard", None)
        self._dist = None
        if shard in ("row", True):
            from .dist import DistContext
            self._dist = DistContext()
            # one rank: nothing to exchange — unless FX_SHARD_WORLD1=1 keeps the whole exchange path
            # alive (ids / rows / gradients all-to-all with itself, all-reduce of one), which is
            # how the RCCL code path is exercised end to end on a 1-GPU box
            if self._dist.world == 1 and os.environ.get("FX_SHARD_WORLD1") != "1":
                self._dist = None
        elif shard not in (None, False, "none"):
            raise ValueError("shard={} is not supported.".format(shard))
        layers.set_dist_context(self._dist)      # tables built below are row-sharded over ranks
        layers.set_emb_dtype(kwargs.get("emb_dtype", "fp32"))   # storage of the D > 1 tables
        # training-control settings, kept under the attribute names model code may read
        self._monitor = Monitor(kv=monitor)
        for attr, value in (("_monitor_mode", monitor_mode), ("_save_best_only", save_best_only),
                            ("_early_stop_patience", early_stop_patience),
                            ("_eval_steps", eval_steps), ("_net_regularizer", net_regularizer),
                            ("_embedding_regularizer", embedding_regularizer),
                            ("_reduce_lr_on_plateau", reduce_lr_on_plateau),
                            ("_verbose", kwargs["verbose"]), ("_max_gradient_norm", 10.),
                            ("_sparse_update", kwargs.get("sparse_update", "exact")),
                            ("_use_graph", bool(kwargs.get("hip_graph", False))),
                            ("_device_metrics", bool(kwargs.get("device_metrics", True))),
                            ("_graph_state", None), ("_graph_warm", 0)):
            setattr(self, attr, value)
        self.feature_map, self.model_id = feature_map, model_id
        self.validation_metrics = kwargs["metrics"]
        self.output_activation = self.get_output_activation(task)
        self.model_dir = os.path.join(kwargs["model_root"], feature_map.dataset_id)
        self.checkpoint = os.path.abspath(os.path.join(self.model_dir, model_id + ".model"))

    def compile(self, optimizer, loss, lr):
        self.optimizer = get_optimizer(optimizer, self.parameters(), lr, model=self,
                                       sparse_update=self._sparse_update,
                                       emb_reg=get_regularizer(self._embedding_regularizer)
                                       if self._embedding_regularizer else None)
        self.loss_fn = get_loss(loss)
        layers.link_fusion(self)
        if self._dist is not None and self._dist.world > 1 and self._use_graph \
                and any(isinstance(m, layers.Dice) for m in self.modules()):
            # Dice all-reduces its batch statistics inside the backward pass (autograd thread); the
            # segmented capture can only be cut from the thread that began it: launch eagerly
            logging.info("hip_graph disabled: Dice statistics are all-reduced inside the backward")
            self._use_graph = False

    def regularization_loss(self):
        """rank_model.py:95-118.  The embedding part is computed by fx_reg_stats and carries no
        autograd graph: its gradient (every table row, every step) is applied inside the native
        update kernels; the net part is plain torch on the dense parameters."""
        total = 0
        if self._embedding_regularizer and hasattr(self.optimizer, "emb_reg_loss"):
            total = total + self.optimizer.emb_reg_loss()
            # parameters of a FeatureEmbeddingDict that are neither packed tables nor numeric weights
            # (feature_encoders: the nn.Linear projection of `embedding`-type features, custom
            # encoders) get the reference's term through autograd (rank_model.py:104-112)
            pairs = get_regularizer(self._embedding_regularizer)
            for mod in self.modules():
                if type(mod) != FeatureEmbeddingDict:
                    continue
                native = {id(p) for p in mod.table_parameters() + mod.numeric_parameters()}
                for weight in mod.parameters():
                    if id(weight) in native or not weight.requires_grad:
                        continue
                    for order, lam in pairs:
                        total = total + (lam / order) * torch.norm(weight, order) ** order
        if not self._net_regularizer:
            return total
        # everything that is not a parameter of a FeatureEmbeddingDict module is a "net" parameter
        table_side = {prefix + "." + pname
                      for prefix, mod in self.named_modules() if type(mod) == FeatureEmbeddingDict
                      for pname, _ in mod.named_parameters()}
        pairs = get_regularizer(self._net_regularizer)
        for pname, weight in self.named_parameters():
            if pname in table_side or not weight.requires_grad:
                continue
            for order, lam in pairs:
                total = total + (lam / order) * torch.norm(weight, order) ** order
        return total

    def add_loss(self, return_dict, y_true):
        return self.loss_fn(return_dict["y_pred"], y_true, reduction='mean')

    def compute_loss(self, return_dict, y_true):
        loss, reg = self.add_loss(return_dict, y_true), self.regularization_loss()
        if isinstance(reg, int) and reg == 0:
            return loss               # no regularizer: no "loss + 0" launch
        return loss + reg

    def reset_parameters(self):
        """rank_model.py:146-167: xavier_normal_ on Linear/Conv1d, then every init_weights()."""
        dense_kinds = (nn.Linear, nn.Conv1d, FxLinear, _NumericView)
        for mod in self.modules():                 # pass 1: Xavier on every dense weight
            if type(mod) in dense_kinds:
                nn.init.xavier_normal_(mod.weight)
                if mod.bias is not None:
                    mod.bias.data.zero_()
        for mod in self.modules():                 # pass 2: layers with their own initializer
            init = getattr(mod, "init_weights", None)
            if callable(init):
                init()

    def get_inputs(self, inputs, feature_source=None):
        """rank_model.py:169-189; returns a FeatureDict so the embedding layers of this model can
        share the packed id matrix and the de-dup of the batch.  Host tensors (what a DataLoader
        yields) are not copied one by one as the reference does (40 small blocking H2D copies per
        batch): all columns of one dtype, labels included, go through one pinned staging buffer and
        one asynchronous copy."""
        if isinstance(inputs, FeatureDict) and getattr(inputs, "_fx_ready", False):
            return inputs
        specs = self.feature_map.features
        names = [name for name in inputs.keys()
                 if name not in self.feature_map.labels and specs[name]["type"] != "meta"
                 and not (feature_source and not_in_whitelist(specs[name]["source"], feature_source))]
        X_dict = FeatureDict()
        staged = self._stage_host_columns(inputs, names)
        for feature in names:
            X_dict[feature] = staged[feature] if feature in staged \
                else inputs[feature].to(self.device)
        return X_dict

    def _stage_host_columns(self, inputs, names):
        """-> {name: device tensor} for the host-resident columns of `names` + the label."""
        self._staged_labels = (None, None)
        if self.device.type != "cuda":
            return {}
        label = self.feature_map.labels[0]
        cols = [f for f in names if not inputs[f].is_cuda]
        if label in inputs and not inputs[label].is_cuda:
            cols = cols + [label]
        if len(cols) < 2:
            return {}
        by_dtype = {}
        for f in cols:
            by_dtype.setdefault(inputs[f].dtype, []).append(f)
        if not hasattr(self, "_pinned"):
            self._pinned, self._pin_turn = {}, 0
        self._pin_turn ^= 1                      # two buffer sets: batch i+1 is staged while the
        out = {}                                 # copy of batch i may still be in flight
        stream = torch.cuda.current_stream(self.device)
        for dtype, feats in by_dtype.items():
            n = sum(inputs[f].numel() for f in feats)
            key = (dtype, self._pin_turn)
            slot = self._pinned.get(key)
            if slot is None or slot[0].numel() < n:
                pinned = torch.empty(n, dtype=dtype, pin_memory=True)
                slot = [pinned, None, pinned.numpy()]
                self._pinned[key] = slot
            if slot[1] is not None:
                slot[1].synchronize()            # the copy that last read this buffer is done
            # gather with numpy straight into the pinned block: measured 0.03 ms for 26 x 4096
            # int64, where torch.cat(out=pinned) / pinned.copy_() take ~3 ms and a pageable
            # .to(device) of the concatenation 1.3 ms (scripts/ubench/stage_probe2.py)
            np.concatenate([inputs[f].detach().numpy().reshape(-1) for f in feats],
                           out=slot[2][:n])
            dev = slot[0][:n].to(self.device, non_blocking=True)
            slot[1] = torch.cuda.Event()
            slot[1].record(stream)
            off = 0
            for f in feats:
                k = inputs[f].numel()
                out[f] = dev[off:off + k].view(inputs[f].shape)
                off += k
        if label in out:
            self._staged_labels = (id(inputs), out.pop(label))
        return out

    def get_labels(self, inputs):
        labels = self.feature_map.labels
        key, staged = getattr(self, "_staged_labels", (None, None))
        y = staged if key == id(inputs) else inputs[labels[0]].to(self.device)
        return y.float().view(-1, 1)

    def get_group_id(self, inputs):
        return inputs[self.feature_map.group_id]

    def model_to_device(self):
        self.to(device=self.device)

    def lr_decay(self, factor=0.1, min_lr=1e-6):
        """rank_model.py:221-234 (the native optimizer picks the new value up in sync_lr)."""
        new_lr = None
        for group in self.optimizer.param_groups:
            new_lr = group["lr"] = max(min_lr, factor * group["lr"])
        return new_lr

    def fit(self, data_generator, epochs=1, validation_data=None,
            max_gradient_norm=10., **kwargs):
        """rank_model.py:236-270: train for `epochs`, evaluate every `eval_steps` steps, keep the
        best checkpoint, stop early on a plateau, reload the best weights at the end."""
        self.valid_gen = validation_data
        self._max_gradient_norm = max_gradient_norm
        self._steps_per_epoch = len(data_generator)
        if self._dist is not None:
            # each rank feeds ITS shard of the data; the step count must agree (collectives)
            self._dist.require_same(self._steps_per_epoch, "number of training batches per epoch")
        if self._eval_steps is None:
            self._eval_steps = self._steps_per_epoch
        maximise = self._monitor_mode != "min"
        self._best_metric = -np.inf if maximise else np.inf
        self._stopping_steps = self._total_steps = 0
        self._batch_index = self._epoch_index = 0
        self._stop_training = False
        logging.info("fit: %d epochs x %d batches, evaluation every %d steps", epochs,
                     self._steps_per_epoch, self._eval_steps)
        for self._epoch_index in range(epochs):
            logging.info("epoch %d begins", self._epoch_index + 1)
            self.train_epoch(data_generator)
            if self._stop_training:
                break
        logging.info("fit done; restoring the best weights from %s", self.checkpoint)
        self.load_weights(self.checkpoint)

    def checkpoint_and_earlystop(self, logs, min_delta=1e-6):
        """rank_model.py:272-298: an evaluation counts as an improvement when the monitored value
        beats the best one by at least `min_delta`; otherwise patience is consumed (and the
        learning rate reduced); the checkpoint follows `save_best_only`."""
        value = self._monitor.get_value(logs)
        sign = -1.0 if self._monitor_mode == "min" else 1.0
        improved = sign * (value - self._best_metric) >= min_delta
        if improved:
            self._best_metric, self._stopping_steps = value, 0
        else:
            self._stopping_steps += 1
            logging.info("no improvement: monitor(%s) = %.6f (best %.6f)", self._monitor_mode, value,
                         self._best_metric)
            if self._reduce_lr_on_plateau:
                logging.info("learning rate reduced to %.6g", self.lr_decay())
        if improved or not self._save_best_only:
            if improved:
                logging.info("new best monitor(%s) = %.6f, saving", self._monitor_mode, value)
            self.save_weights(self.checkpoint)
        if self._stopping_steps >= self._early_stop_patience:
            self._stop_training = True
            logging.info("early stop in epoch %d", self._epoch_index + 1)

    def eval_step(self):
        logging.info("evaluation at epoch %d, batch %d", self._epoch_index + 1, self._batch_index + 1)
        self.checkpoint_and_earlystop(self.evaluate(self.valid_gen,
                                                    metrics=self._monitor.get_metrics()))
        self.train()

    def train(self, mode=True):
        """Leaving training mode makes the tables consistent first: in exact mode rows that were
        not read recently still owe their zero-gradient Adam steps (optim.py)."""
        if not mode and getattr(self, "optimizer", None) is not None \
                and hasattr(self.optimizer, "flush"):
            self.optimizer.flush()
        return super().train(mode)

    def _forward_backward(self, batch_data):
        """zero_grad; forward; loss; backward (rank_model.py:308-320) -> loss.  Afterwards the dense
        gradients sit in .grad, the table gradients (unique rows) in the table groups' `pending`."""
        opt = self.optimizer
        opt.zero_grad()                  # also opens the step: t += 1, Adam bias corrections
        act = self.output_activation
        fused = (isinstance(act, FxSigmoid) and self.loss_fn is _bce_loss
                 and type(self).add_loss is BaseModel.add_loss)
        # the root gradient of this step: 1, or 1 / world when the batch is one rank's share (global-batch
        # mean = mean of the ranks' local means: scale, then SUM-reduce the gradients)
        if self._dist is not None and self._dist.world > 1:
            root, root_scale = _scaled_grad(self.device, self._dist.world), 1.0 / self._dist.world
        else:
            root, root_scale = _unit_grad(self.device), 1.0
        hc = None
        if fused:
            act.defer = True        # nobody reads the probabilities of a training step
            if layers._HEAD_FUSED:
                # the labels are part of the batch: a tower ending in Linear(K -> 1) may evaluate head +
                # loss + head backward in one pass (layers._HeadCtx; it asks for the labels when it gets
                # there, after get_inputs() staged the batch)
                hc = layers._HEAD_CTX = layers._HeadCtx(lambda: self.get_labels(batch_data), root_scale,
                                                        root.data_ptr())
        try:
            return_dict = self.forward(batch_data)
            y_true = hc.labels_taken if (hc is not None and hc.labels_taken is not None) \
                else self.get_labels(batch_data)
            loss = self.compute_loss(return_dict, y_true)
        finally:
            if fused:
                act.defer = False
                layers._HEAD_CTX = None
                layers._RELU_NOTES.clear()
        loss.backward(gradient=root)
        return loss

    def _step_body(self, batch_data):
        loss = self._forward_backward(batch_data)
        self.optimizer.step()  # global-norm clip (rank_model.py:321) is fused into the update kernels
        return loss

    def load_full_state_dict(self, full_state):
        """Load a FULL (reference-layout) state dict into a row-sharded model: every rank keeps
        its rows of each table; dense parameters are copied as they are."""
        own = self.state_dict()
        dense = {k: v for k, v in full_state.items() if k in own and own[k].shape == v.shape
                 and ".embedding_layers." not in k}
        self.load_state_dict(dense, strict=False)
        for name, mod in self.named_modules():
            if isinstance(mod, FeatureEmbeddingDict):
                mod.load_full_tables(full_state, prefix=name + ".")

    def full_state_dict(self):
        """FULL (reference-layout) state dict of a row-sharded model (tables are all-gathered)."""
        if hasattr(self.optimizer, "flush"):
            self.optimizer.flush()
        out = {k: v for k, v in self.state_dict().items() if ".embedding_layers." not in k}
        for name, mod in self.named_modules():
            if isinstance(mod, FeatureEmbeddingDict):
                out.update(mod.gather_full_tables(prefix=name + "."))
        return out

    def train_step(self, batch_data):
        """rank_model.py:307-323 on the native path.  With `hip_graph: true` the whole step
        (about 60 launches) is captured once into a hipGraph and replayed: every per-step scalar
        (step counter, lr, clip coefficient, unique-row count) already lives in device memory."""
        opt = self.optimizer
        if not getattr(opt, "_max_norm_explicit", False):
            opt.set_max_norm(self._max_gradient_norm, _from_model=True)
        opt.sync_lr()
        if self._use_graph:
            return self._train_step_graph(batch_data)
        return self._step_body(batch_data)

    # -- hipGraph replay of the training step ----------------------------------------------------
    def _side_stream_step(self, batch_data):
        """Warm-up steps of graph mode run on the stream the capture will use, so autograd's
        AccumulateGrad nodes (created at the first backward, remembered per parameter) are not
        tied to the default stream — a capture may not depend on it."""
        if getattr(self, "_graph_stream", None) is None:
            self._graph_stream = torch.cuda.Stream(self.device)
        s = self._graph_stream
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            loss = self._step_body(batch_data)
        torch.cuda.current_stream(self.device).wait_stream(s)
        return loss

    def _train_step_graph(self, batch_data):
        st = self._graph_state
        label = self.feature_map.labels[0]
        B = batch_data[label].shape[0]
        if st is None:
            if self._graph_warm < 3:          # eager warm-up: allocations, plans, workspaces
                self._graph_warm += 1
                return self._side_stream_step(batch_data)
            st = self._graph_state = _GraphStep(self, batch_data)
            return st.probe_loss              # the probe inside _GraphStep WAS this batch's step
        if B != st.B:
            # e.g. the last, shorter batch of an epoch: eager, on the stream the parameters'
            # AccumulateGrad nodes were created on
            return self._side_stream_step(batch_data)
        st.fill(batch_data)
        st.graph.replay()
        return st.loss

    def release_graphs(self):
        """Destroy the captured step (and with it every recorded RCCL kernel's hold on the
        communicator); the next train_step captures again."""
        self._graph_state = None
        self._graph_warm = 0

    def prepare_batch(self, batch_data):
        """Optional: do the host side of a captured step's input cast for a device-resident batch
        ahead of time (the first train_step on a batch object does it otherwise).  No launch, no
        effect on results; a no-op before the step has been captured or for host batches."""
        st = self._graph_state
        if (st is not None and type(batch_data) is dict
                and all(getattr(v, "is_cuda", False) for v in batch_data.values())
                and batch_data[self.feature_map.labels[0]].shape[0] == st.B):
            st.fill(batch_data, launch=False, registered=True)

    def _progress(self, iterable):
        if self._verbose > 0:
            from tqdm import tqdm
            return tqdm(iterable, disable=False, file=sys.stdout)
        return iterable

    def train_epoch(self, data_generator):
        """rank_model.py:325-348."""
        self.train()
        # The reference adds `loss.item()` to a running sum every step (rank_model.py:333): a host round
        # trip that drains the device before the next batch is even staged.  The sum is only ever LOGGED
        # (every eval_steps), so it is kept on the device (float64, one tiny add behind each step) and
        # read when it is printed; the host runs a step ahead of the GPU in between.
        window_loss, self._batch_index = None, 0
        for self._batch_index, batch in enumerate(self._progress(data_generator)):
            loss = self.train_step(batch)
            if window_loss is None:
                window_loss = torch.zeros((), dtype=torch.float64, device=loss.device)
            window_loss.add_(loss.detach())
            self._total_steps += 1
            if self._total_steps % self._eval_steps == 0:
                logging.info("mean train loss over the last %d steps: %.6f", self._eval_steps,
                             float(window_loss.item()) / self._eval_steps)
                window_loss.zero_()
                self.eval_step()
            if self._stop_training:
                break
        self.optimizer.check_errors()

    def _predict_batches(self, data_generator, with_labels):
        """eval-mode forward over a generator -> (list of device predictions, list of labels)."""
        self.eval()
        preds, labels = [], []
        with torch.no_grad():
            for batch in self._progress(data_generator):
                preds.append(self.forward(batch)["y_pred"].detach().reshape(-1).float())
                if with_labels:
                    # clone: a device loader may hand out views of a buffer it reuses
                    labels.append(self.get_labels(batch).reshape(-1).clone())
        return preds, labels

    def evaluate(self, data_generator, metrics=None):
        """rank_model.py:350-381.  logloss / AUC are computed on the device (fx_binary_metrics: one
        sort + exact rank sums over all predictions) unless group metrics are involved, in which
        case the reference's host path (float64 + scikit-learn) is used."""
        wanted = list(self.validation_metrics if metrics is None else metrics)
        groups = self.feature_map.group_id is not None
        on_device = (self._device_metrics and self.device.type == "cuda" and not groups and wanted
                     and set(wanted) <= {"logloss", "binary_crossentropy", "AUC"})
        if self._dist is not None:
            # every rank scores the GLOBAL validation set (its shard's predictions are gathered),
            # so checkpoint_and_earlystop takes the same decision everywhere
            self._dist.require_same(len(data_generator), "number of evaluation batches")
        if on_device:
            preds, labels = self._predict_batches(data_generator, True)
            preds, labels = torch.cat(preds), torch.cat(labels)
            if self._dist is not None:
                preds, labels = (self._dist.all_gather_cat(t) for t in (preds, labels))
            ll, auc = ops.binary_metrics(preds, labels)
            val_logs = OrderedDict((m, auc if m == "AUC" else ll) for m in wanted)
        else:
            self.eval()
            p_host, y_host, g_host = [], [], []
            with torch.no_grad():
                for batch in self._progress(data_generator):
                    p_host.append(self.forward(batch)["y_pred"].detach().reshape(-1).cpu().numpy())
                    y_host.append(self.get_labels(batch).reshape(-1).cpu().numpy())
                    if groups:
                        g_host.append(np.asarray(self.get_group_id(batch)).reshape(-1))
            y_all, p_all = np.concatenate(y_host), np.concatenate(p_host)
            g_all = np.concatenate(g_host) if g_host else None
            if self._dist is not None:
                y_all, p_all = (self._dist.all_gather_cat(torch.from_numpy(a)).numpy()
                                for a in (y_all, p_all))
                if g_all is not None:
                    g_all = self._dist.all_gather_cat(torch.from_numpy(g_all)).numpy()
            val_logs = self.evaluate_metrics(y_all.astype(np.float64), p_all.astype(np.float64),
                                             wanted, g_all)
        logging.info("metrics: %s", ", ".join("%s=%.6f" % kv for kv in val_logs.items()))
        return val_logs

    def predict(self, data_generator):
        """rank_model.py:383-398 -> float64 numpy vector of probabilities."""
        preds, _ = self._predict_batches(data_generator, False)
        return torch.cat(preds).cpu().numpy().astype(np.float64)

    def evaluate_metrics(self, y_true, y_pred, metrics, group_id=None):
        return evaluate_metrics(y_true, y_pred, metrics, group_id)

    def _shard_path(self, checkpoint):
        """Row-sharded tables: every rank owns a file with ITS rows (+ the replicated dense part)."""
        if self._dist is None:
            return checkpoint
        return "%s.rank%d-of-%d" % (checkpoint, self._dist.rank, self._dist.world)

    def save_weights(self, checkpoint):
        if hasattr(self.optimizer, "flush"):
            self.optimizer.flush()
        os.makedirs(os.path.dirname(checkpoint), exist_ok=True)
        # (a table inside a row record is a strided view: saved as such, torch would write the record's
        # whole storage — moments included — so those entries are compacted first)
        state = OrderedDict((k, v if v.is_contiguous() else v.contiguous())
                            for k, v in self.state_dict().items())
        torch.save(state, self._shard_path(checkpoint))

    def load_weights(self, checkpoint):
        self.to(self.device)
        if getattr(self, "optimizer", None) is not None and hasattr(self.optimizer, "flush"):
            # exact mode: settle every row's pending zero-gradient replays BEFORE the tables are
            # overwritten — otherwise they would be applied on top of the loaded weights at the next
            # flush, and e.g. fit()'s final evaluation would not see exactly the saved checkpoint
            self.optimizer.flush()
        path = self._shard_path(checkpoint)
        if self._dist is not None and not os.path.exists(path):
            # a full (reference-layout) checkpoint: every rank picks its rows out of it
            self.load_full_state_dict(torch.load(checkpoint, map_location="cpu"))
            return
        state_dict = torch.load(path, map_location="cpu")
        self.load_state_dict(state_dict)

    def save_checkpoint(self, checkpoint):
        """Weights + optimizer state (moments, row stamps, step, lr): a true resume point, which the
        reference does not have (it saves weights only, rank_model.py:423)."""
        self.save_weights(checkpoint)
        torch.save(self.optimizer.state_dict(), self._shard_path(checkpoint) + ".optim")

    def load_checkpoint(self, checkpoint):
        self.load_weights(checkpoint)
        self.optimizer.load_state_dict(torch.load(self._shard_path(checkpoint) + ".optim",
                                                  map_location="cpu", weights_only=False))
        self._graph_state = None          # a captured step holds the old step's buffers only

    def get_output_activation(self, task):
        if task == "binary_classification":
            return FxSigmoid()
        elif task == "regression":
            return nn.Identity()
        else:
            raise NotImplementedError("task={} is not supported.".format(task))

    def count_parameters(self, count_embedding=True):
        """rank_model.py:454-470."""
        n = sum(w.numel() for pname, w in self.named_parameters()
                if w.requires_grad and (count_embedding or "embedding" not in pname))
        logging.info("trainable parameters: %d", n)
        return n
