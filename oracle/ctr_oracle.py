"""CPU oracle: a torch-fp32 restatement of the reference hot path.

*** TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this file; the product (fuxictr_amd/) never does. ***

Where the arithmetic lives: the reference (reczoo/FuxiCTR v2.3.10) is pure Python on top of
PyTorch (not vendored, unpinned: requirements.txt lists no torch; README.md:112 says >=2.7.1), so
its numerics ARE the ATen CPU kernels.  This file restates the reference's algorithm as explicit
functional torch-CPU code — per-field lookups, stack, LR, FM, MLP, CrossNetV2, sigmoid+BCE,
clip_grad_norm_, dense Adam — each function citing the reference lines it follows (paths
relative to the reference checkout).  It deliberately keeps the reference's cost structure (dense
[V, D] gradients, dense Adam over every row), so timing `train_step` is a fair "port" baseline.

Pinned against the real reference: tests/golden/*.npz were produced by importing the actual
reference modules from /root/reference in the build container (tests/golden/make_golden.py) and
tests/test_oracle_golden.py checks every function here against them.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

EMB = "embedding_layer.embedding_layer.embedding_layers."
LR_EMB = "fm.lr_layer.embedding_layer.embedding_layer.embedding_layers."


# ---------------------------------------------------------------------------------------------
# layers
# ---------------------------------------------------------------------------------------------
def masked_average_pooling(embedding_matrix):
    """MaskedAveragePooling.forward, pooling.py:32-47 with mask=None: a position counts when its
    embedding row does not sum to exactly 0 (the padding row); +1e-12 keeps empty histories finite."""
    sum_out = torch.sum(embedding_matrix, dim=1)
    mask = embedding_matrix.sum(dim=-1) != 0
    return sum_out / (mask.float().sum(-1, keepdim=True) + 1.e-12)


def masked_sum_pooling(embedding_matrix):
    """MaskedSumPooling.forward, pooling.py:59-70 (padding rows are zero vectors)."""
    return torch.sum(embedding_matrix, dim=1)


ENCODERS = {"layers.MaskedAveragePooling()": masked_average_pooling,
            "layers.MaskedSumPooling()": masked_sum_pooling}


def feature_embedding(state, prefix, features, X, encode=True):
    """FeatureEmbeddingDict.forward, feature_embedding.py:261-297 -> OrderedDict name -> tensor.
    numeric: x.float().view(-1,1) @ W[D,1]^T (:280-282); categorical/sequence: W[ids.long()]
    (:283-288), then the feature's `feature_encoder` (:294-295) — encode=False for the D=1 LR copy,
    which installs its own sum pooling instead (:135-138).  `share_embedding` aliases are resolved
    by the state_dict (both keys exist)."""
    out = OrderedDict()
    for name, spec in features.items():
        if name not in X or spec["type"] == "meta":
            continue
        w = state[prefix + name + ".weight"]
        if spec["type"] == "numeric":
            # x.float() in the reference; .to(w.dtype) is the same in fp32 and lets OracleTrainer64
            # evaluate the identical graph in float64
            out[name] = F.linear(X[name].to(w.dtype).view(-1, 1), w)
        elif spec["type"] in ("categorical", "sequence"):
            out[name] = F.embedding(X[name].long(), w, padding_idx=spec.get("padding_idx", None))
            if encode and spec.get("feature_encoder"):
                out[name] = ENCODERS[spec["feature_encoder"]](out[name])
        else:
            raise NotImplementedError(spec["type"])
    return out


def dict2tensor(features, emb, flatten_emb=False):
    """FeatureEmbeddingDict.dict2tensor, feature_embedding.py:230-259 (no filters)."""
    lst = [emb[f] for f in features if f in emb]
    return torch.cat(lst, dim=-1) if flatten_emb else torch.stack(lst, dim=1)


def logistic_regression(state, features, X, emb_prefix=LR_EMB, bias_key="fm.lr_layer.bias"):
    """LogisticRegression.forward, logistic_regression.py:46-59: a D=1 FeatureEmbedding summed over
    fields + bias (sequence features are sum-pooled first, feature_embedding.py:135-138)."""
    emb = feature_embedding(state, emb_prefix, features, X, encode=False)
    for name, spec in features.items():
        if name in emb and spec["type"] == "sequence":
            emb[name] = emb[name].sum(dim=1)
    w = dict2tensor(features, emb)       # [B, F, 1]
    out = w.sum(dim=1)
    if bias_key in state:
        out = out + state[bias_key]
    return out


def fm_product_sum(feature_emb):
    """InnerProductInteraction 'product_sum', inner_product.py:55-62."""
    sum_of_square = torch.sum(feature_emb, dim=1) ** 2
    square_of_sum = torch.sum(feature_emb ** 2, dim=1)
    return ((sum_of_square - square_of_sum) * 0.5).sum(dim=-1, keepdim=True)


def mlp_block(state, prefix, x, n_hidden, has_output, batch_norm=False, training=False):
    """MLP_Block.forward, mlp_block.py:53-96 for Linear[+BatchNorm1d]+ReLU stacks (no dropout): in
    the nn.Sequential a hidden layer occupies 2 slots (Linear, ReLU) or 3 (Linear, BN, ReLU); the
    output Linear follows.  BatchNorm1d defaults: eps 1e-5, momentum 0.1, batch statistics in
    training (running statistics and num_batches_tracked updated in place), running ones in eval."""
    h = x
    per = 3 if batch_norm else 2
    for i in range(n_hidden):
        h = F.linear(h, state[prefix + "mlp.%d.weight" % (per * i)],
                     state[prefix + "mlp.%d.bias" % (per * i)])
        if batch_norm:
            bn = prefix + "mlp.%d." % (per * i + 1)
            h = F.batch_norm(h, state[bn + "running_mean"], state[bn + "running_var"],
                             state[bn + "weight"], state[bn + "bias"], training, 0.1, 1e-5)
            if training:
                state[bn + "num_batches_tracked"] += 1
        h = F.relu(h)
    if has_output:
        k = per * n_hidden
        h = F.linear(h, state[prefix + "mlp.%d.weight" % k], state[prefix + "mlp.%d.bias" % k])
    return h


def crossnet_v2(state, prefix, x0, n_layers):
    """CrossNetV2.forward, cross_net.py:117-129: X_{i+1} = X_i + X_0 * (W_i X_i + b_i)."""
    xi = x0
    for i in range(n_layers):
        xi = xi + x0 * F.linear(xi, state[prefix + "cross_layers.%d.weight" % i],
                                state[prefix + "cross_layers.%d.bias" % i])
    return xi


# ---------------------------------------------------------------------------------------------
# models (logit before the output activation)
# ---------------------------------------------------------------------------------------------
def deepfm_logit(state, features, X, n_hidden, batch_norm=False, training=False):
    """DeepFM.forward, model_zoo/DeepFM/DeepFM_torch/src/DeepFM.py:73-88."""
    emb = dict2tensor(features, feature_embedding(state, EMB, features, X))
    y = fm_product_sum(emb) + logistic_regression(state, features, X)   # factorization_machine.py:46-59
    return y + mlp_block(state, "mlp.", emb.flatten(start_dim=1), n_hidden, True, batch_norm,
                         training)


def dcnv2_logit(state, features, X, n_cross, n_hidden, structure="parallel", n_stacked=0):
    """DCNv2.forward, model_zoo/DCNv2/src/DCNv2.py:108-132: the cross network on the flattened
    embeddings, then per model_structure — alone, through the stacked DNN, beside the parallel DNN,
    or both — into the final Linear.  The two DNNs end on their last hidden layer (output_dim=None)."""
    emb = dict2tensor(features, feature_embedding(state, EMB, features, X), flatten_emb=True)
    cross = crossnet_v2(state, "crossnet.", emb, n_cross)
    if structure == "crossnet_only":
        final = cross
    elif structure == "stacked":
        final = mlp_block(state, "stacked_dnn.", cross, n_stacked, False)
    elif structure == "parallel":
        final = torch.cat([cross, mlp_block(state, "parallel_dnn.", emb, n_hidden, False)], dim=-1)
    elif structure == "stacked_parallel":
        final = torch.cat([mlp_block(state, "stacked_dnn.", cross, n_stacked, False),
                           mlp_block(state, "parallel_dnn.", emb, n_hidden, False)], dim=-1)
    else:
        raise ValueError("model_structure={} not supported!".format(structure))
    return F.linear(final, state["fc.weight"], state["fc.bias"])


def dice(state, prefix, x, training):
    """Dice, fuxictr/pytorch/layers/activations.py:40-51: BatchNorm1d(affine=False, eps=1e-9,
    momentum=0.01) over ALL rows (training: batch statistics, running statistics updated in place),
    then p = sigmoid(.), y = p x + alpha (1-p) x."""
    bn = F.batch_norm(x, state[prefix + "bn.running_mean"], state[prefix + "bn.running_var"],
                      None, None, training, 0.01, 1e-9)
    if training:
        state[prefix + "bn.num_batches_tracked"] += 1
    p = torch.sigmoid(bn)
    return p * x + state[prefix + "alpha"] * (1 - p) * x


def din_attention(state, prefix, target, seq, mask, training, use_softmax=False):
    """DIN_Attention.forward, target_attention.py:66-92; the attention MLP is
    Linear(4E,H) -> Dice(H) -> Linear(H,1) (mlp.0 / mlp.1 / mlp.2).  use_softmax: masked positions
    get -1e9 before a softmax over the history (:86-89)."""
    L = seq.size(1)
    E = target.size(-1)
    t = target.unsqueeze(1).expand(-1, L, -1)
    x = torch.cat([t, seq, t - seq, t * seq], dim=-1).view(-1, 4 * E)
    h = F.linear(x, state[prefix + "attention_layer.mlp.0.weight"],
                 state[prefix + "attention_layer.mlp.0.bias"])
    h = dice(state, prefix + "attention_layer.mlp.1.", h, training)
    w = F.linear(h, state[prefix + "attention_layer.mlp.2.weight"],
                 state[prefix + "attention_layer.mlp.2.bias"]).view(-1, L)
    w = w * mask.float()
    if use_softmax:
        w = (w + -1.e9 * (1 - mask.float())).softmax(dim=-1)
    return (w.unsqueeze(-1) * seq).sum(dim=1)


DIN_EMB = "embedding_layer.embedding_layers."


def din_logit(state, features, X, cfg, training):
    """DIN.forward, model_zoo/DIN/src/DIN.py:109-133.  A target / sequence entry is a field name or
    a tuple of names: tuples are concatenated on the last dim (get_embedding, :135-148), the mask
    comes from the FIRST sequence field, and the pooled vector is split back into embedding_dim-wide
    pieces that replace the sequence entries."""
    emb = feature_embedding(state, DIN_EMB, features, X)

    def fields(f):
        return list(f) if isinstance(f, (tuple, list)) else [f]

    for idx, (tf, sf) in enumerate(zip(cfg["din_target_field"], cfg["din_sequence_field"])):
        target = torch.cat([emb[f] for f in fields(tf)], dim=-1)
        seq = torch.cat([emb[f] for f in fields(sf)], dim=-1)
        mask = X[fields(sf)[0]].long() != 0
        pooled = din_attention(state, "attention_layers.%d." % idx, target, seq, mask, training,
                               cfg.get("din_softmax", False))
        if len(fields(sf)) == 1:
            emb[fields(sf)[0]] = pooled
        else:
            for f, piece in zip(fields(sf), pooled.split(cfg["embedding_dim"], dim=-1)):
                emb[f] = piece
    x = dict2tensor(features, emb, flatten_emb=True)
    return mlp_block(state, "dnn.", x, cfg["n_hidden"], True)


def dot_interaction(feature_emb):
    """InnerProductInteraction 'inner_product', inner_product.py:63-66."""
    F_ = feature_emb.shape[1]
    ipm = torch.bmm(feature_emb, feature_emb.transpose(1, 2))
    mask = torch.triu(torch.ones(F_, F_, device=feature_emb.device), 1).bool()
    return torch.masked_select(ipm, mask).view(-1, F_ * (F_ - 1) // 2)


def dlrm_logit(state, features, X, cfg):
    """DLRM.forward (interaction_op 'dot' or 'cat'), model_zoo/DLRM/src/DLRM.py:103-123: sparse-only
    FeatureEmbedding, bottom MLP (ReLU after its last layer too) on the numeric columns appended as
    an extra field, pairwise dots ++ dense embedding -> top MLP."""
    sparse = OrderedDict((f, s) for f, s in features.items() if s["type"] != "numeric")
    dense_feats = [f for f, s in features.items() if s["type"] == "numeric"]
    emb = dict2tensor(sparse, feature_embedding(state, EMB, sparse, X))
    if dense_feats:
        dense_x = torch.cat([X[k].float().view(-1, 1) for k in dense_feats], dim=-1)
        h = dense_x.to(state["bottom_mlp.mlp.0.weight"].dtype)     # (fp32; fp64 in OracleTrainer64)
        for i in range(cfg["n_bottom"] + 1):
            h = F.relu(F.linear(h, state["bottom_mlp.mlp.%d.weight" % (2 * i)],
                                state["bottom_mlp.mlp.%d.bias" % (2 * i)]))
        emb = torch.cat([emb, h.unsqueeze(1)], dim=1)
    if cfg.get("interaction_op", "dot") == "cat":            # nn.Flatten(start_dim=1), DLRM.py:88-89
        inter = emb.flatten(start_dim=1)
    elif dense_feats:
        inter = torch.cat([dot_interaction(emb), h], dim=-1)
    else:
        inter = dot_interaction(emb)
    return mlp_block(state, "top_mlp.", inter, cfg["n_hidden"], True)


def cin(state, prefix, feature_emb, n_layers):
    """CompressedInteractionNet.forward, compressed_interaction_net.py:54-76."""
    X_0 = feature_emb
    B, _, D = X_0.shape
    X_i = X_0
    pooled = []
    for i in range(n_layers):
        had = torch.einsum("bhd,bmd->bhmd", X_0, X_i).view(B, -1, D)
        X_i = F.conv1d(had, state[prefix + "cin_layer.layer_%d.weight" % (i + 1)],
                       state[prefix + "cin_layer.layer_%d.bias" % (i + 1)]).view(B, -1, D)
        pooled.append(X_i.sum(dim=-1))
    return F.linear(torch.cat(pooled, dim=-1), state[prefix + "fc.weight"], state[prefix + "fc.bias"])


def xdeepfm_logit(state, features, X, cfg):
    """xDeepFM.forward, model_zoo/xDeepFM/src/xDeepFM.py:78-97 (LR without bias + CIN + DNN)."""
    emb = dict2tensor(features, feature_embedding(state, EMB, features, X))
    lr = logistic_regression(state, features, X,
                             emb_prefix="lr_layer.embedding_layer.embedding_layer.embedding_layers.",
                             bias_key="lr_layer.bias")
    y = lr + cin(state, "cin.", emb, cfg["n_cin"])
    return y + mlp_block(state, "dnn.", emb.flatten(start_dim=1), cfg["n_hidden"], True)


def model_logit(cfg, state, features, X, training=False):
    if cfg["model"] == "xDeepFM":
        return xdeepfm_logit(state, features, X, cfg)
    if cfg["model"] == "DLRM":
        return dlrm_logit(state, features, X, cfg)
    if cfg["model"] == "DIN":
        return din_logit(state, features, X, cfg, training)
    if cfg["model"] == "DeepFM":
        return deepfm_logit(state, features, X, cfg["n_hidden"], cfg.get("batch_norm", False),
                            training)
    if cfg["model"] == "DCNv2":
        return dcnv2_logit(state, features, X, cfg["n_cross"], cfg["n_hidden"],
                           cfg.get("structure", "parallel"), cfg.get("n_stacked", 0))
    raise NotImplementedError(cfg["model"])


def predict(cfg, state, features, X):
    """y_pred = sigmoid(logit), rank_model.py:447-448."""
    with torch.no_grad():
        return torch.sigmoid(model_logit(cfg, state, features, X))


# ---------------------------------------------------------------------------------------------
# training step with the reference's dense semantics
# ---------------------------------------------------------------------------------------------
def bce_mean(y_pred, y_true):
    """F.binary_cross_entropy(..., reduction='mean'), torch_utils.py:95-98 / rank_model.py:130."""
    return F.binary_cross_entropy(y_pred, y_true, reduction="mean")


def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ (rank_model.py:321): total = || [ ||g_i||_2 ] ||_2,
    coef = clamp(max_norm / (total + 1e-6), max=1); grads scaled in place.  Returns total."""
    norms = [torch.linalg.vector_norm(g, 2) for g in grads]
    total = torch.linalg.vector_norm(torch.stack(norms), 2)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return float(total)


def adam_dense(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam._single_tensor_adam (the optimizer rank_model.py:322 steps; defaults from
    torch_utils.py:72-76), applied to EVERY element — untouched table rows included."""
    m.lerp_(g, 1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    step_size = lr / bc1
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-step_size)


def parse_regularizer(reg):
    """get_regularizer, torch_utils.py:106-135: float -> L2; "l1(x)", "l2(x)", "l1_l2(x,y)"."""
    if isinstance(reg, float):
        return [(2, reg)]
    if isinstance(reg, str):
        val = reg.rstrip(")").split("(")[-1]
        if reg.startswith("l1(") or reg.startswith("l2("):
            return [(int(reg[1]), float(val))]
        if reg.startswith("l1_l2"):
            a, b = val.split(",")
            return [(1, float(a)), (2, float(b))]
        raise NotImplementedError("regularizer={} is not supported.".format(reg))
    return []


class OracleTrainer(object):
    """State + dense Adam, one `train_step` == BaseModel.train_step (rank_model.py:307-323)."""

    def __init__(self, cfg, state, features, lr=1e-3, max_norm=10.0, optimizer="adam",
                 emb_reg=None, net_reg=None, device="cpu"):
        """device: "cpu" is THE oracle.  A HIP device runs the identical functional code on ATen's
        GPU kernels — what the reference itself executes with `gpu: 0` — and is used by the tests
        only as a yardstick for how far the reference's own two back ends drift apart."""
        self.cfg, self.features = cfg, features
        self.device = torch.device(device)
        self.emb_reg, self.net_reg = emb_reg or [], net_reg or []   # [(p_norm, weight)]
        self.lr, self.max_norm, self.kind = lr, max_norm, optimizer
        self.state = OrderedDict()
        # share_embedding (feature_embedding.py:149-151): in the main table dict the sharing
        # feature's key is the SAME Parameter as its target's key
        alias = {}
        for pre in (EMB, DIN_EMB):
            alias.update({pre + f + ".weight": pre + spec["share_embedding"] + ".weight"
                          for f, spec in features.items() if spec.get("share_embedding")})
        for k, t in state.items():
            if k in alias and alias[k] in self.state:
                self.state[k] = self.state[alias[k]]
                continue
            t = torch.as_tensor(t)
            if "running_" in k or "num_batches_tracked" in k:      # BatchNorm buffers
                self.state[k] = t.detach().clone().to(self.device)
                continue
            self.state[k] = t.detach().clone().float().to(self.device).requires_grad_(True)
        self.params, self.is_emb = [], []
        seen = set()
        for k, t in self.state.items():
            if id(t) not in seen and t.requires_grad:
                seen.add(id(t))
                self.params.append(t)
                # parameters of a FeatureEmbeddingDict module (rank_model.py:106-109)
                self.is_emb.append(".embedding_layers." in k or ".feature_encoders." in k)
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.step = 0

    def _to_dev(self, X):
        if self.device.type == "cpu":
            return X
        return {k: (v.to(self.device) if torch.is_tensor(v) else v) for k, v in X.items()}

    def train_step(self, X, y):
        X, y = self._to_dev(X), y.to(self.device)
        for p in self.params:
            p.grad = None                                   # optimizer.zero_grad()
        prob = torch.sigmoid(model_logit(self.cfg, self.state, self.features, X, training=True))
        loss = bce_mean(prob, y.float().view(-1, 1)) + self.regularization_loss()
        loss.backward()                                     # dense [V, D] embedding grads
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self.params]
        total = clip_grad_norm(grads, self.max_norm)
        self.step += 1
        with torch.no_grad():
            for p, g, m, v in zip(self.params, grads, self.m, self.v):
                if self.kind == "adam":
                    adam_dense(p, g, m, v, self.step, self.lr)
                else:
                    p.add_(g, alpha=-self.lr)               # torch.optim.SGD defaults
        return float(loss.detach()), total

    def gradients(self, X, y, double=False):
        """The gradients loss.backward() (rank_model.py:320) leaves in .grad for batch (X, y) at the
        current weights, BEFORE clip_grad_norm_ — {state key: dense gradient} — evaluated in fp32
        (double=False: what train_step uses) or with the forward / backward in float64 (the yardstick
        of OracleTrainer64).  The weights are not touched."""
        X, y = self._to_dev(X), y.to(self.device)
        state = self.state
        if double:
            twin = {}
            state = OrderedDict()
            for k, t in self.state.items():
                if id(t) not in twin:
                    twin[id(t)] = (t.detach().double().requires_grad_(t.requires_grad)
                                   if t.is_floating_point() else t.clone())
                state[k] = twin[id(t)]
            X = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v)
                 for k, v in X.items()}
            y = y.double()
        else:
            state = OrderedDict((k, (t.detach().clone().requires_grad_(True) if t.requires_grad else
                                     t.clone())) for k, t in self.state.items())
            # share_embedding aliases stay aliases
            first = {}
            for k, t in self.state.items():
                if id(t) in first:
                    state[k] = state[first[id(t)]]
                else:
                    first[id(t)] = k
        prob = torch.sigmoid(model_logit(self.cfg, state, self.features, X, training=True))
        loss = bce_mean(prob, y.to(prob.dtype).view(-1, 1))
        loss.backward()
        out = OrderedDict()
        for k, t in state.items():
            if torch.is_tensor(t) and t.requires_grad:
                out[k] = (t.grad if t.grad is not None else torch.zeros_like(t)).detach().cpu()
        return out

    def regularization_loss(self):
        """BaseModel.regularization_loss, rank_model.py:95-118."""
        reg_term = 0
        for p, emb in zip(self.params, self.is_emb):
            for norm_p, lam in (self.emb_reg if emb else self.net_reg):
                reg_term = reg_term + (lam / norm_p) * torch.norm(p, norm_p) ** norm_p
        return reg_term

    def logits(self, X):
        with torch.no_grad():
            return model_logit(self.cfg, self.state, self.features, self._to_dev(X),
                               training=False).reshape(-1).float().cpu()

    def predict(self, X):
        return predict(self.cfg, self.state, self.features, self._to_dev(X)).cpu()


class OracleTrainer64(OracleTrainer):
    """The SAME algorithm with the forward / backward evaluated in float64: gradients are rounded to
    fp32 once, then the identical fp32 clip + Adam/SGD on the fp32 master weights.  Not a second
    reference — a yardstick for the reference's own conditioning: how far two correctly rounded
    evaluations of one training step drift apart (Adam's lr*m/(sqrt(v)+eps) turns gradient elements
    that are cancellation residue into +-lr steps), which bounds what "equal to the reference after k
    steps" can mean for ANY fp32 implementation (tests/baseline_shapes.py)."""

    def train_step(self, X, y):
        twin, st64 = {}, OrderedDict()
        for k, t in self.state.items():
            if id(t) not in twin:
                if t.is_floating_point():
                    twin[id(t)] = t.detach().double().requires_grad_(t.requires_grad)
                else:
                    twin[id(t)] = t.clone()
            st64[k] = twin[id(t)]
        X64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v)
               for k, v in X.items()}
        self_state, self.state = self.state, st64           # regularization_loss reads self.params
        try:
            prob = torch.sigmoid(model_logit(self.cfg, st64, self.features, X64, training=True))
            loss = bce_mean(prob, y.double().view(-1, 1))
            for p, emb in zip(self.params, self.is_emb):
                for norm_p, lam in (self.emb_reg if emb else self.net_reg):
                    loss = loss + (lam / norm_p) * torch.norm(twin[id(p)], norm_p) ** norm_p
        finally:
            self.state = self_state
        loss.backward()
        grads = [twin[id(p)].grad.float() if twin[id(p)].grad is not None else torch.zeros_like(p)
                 for p in self.params]
        total = clip_grad_norm(grads, self.max_norm)
        self.step += 1
        with torch.no_grad():
            for p, g, m, v in zip(self.params, grads, self.m, self.v):
                if self.kind == "adam":
                    adam_dense(p, g, m, v, self.step, self.lr)
                else:
                    p.add_(g, alpha=-self.lr)
            for k, t in self.state.items():                  # BatchNorm / Dice running statistics
                if not t.requires_grad:
                    t.copy_(st64[k].to(t.dtype))
        return float(loss.detach()), total

