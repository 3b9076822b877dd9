"""Host-logic test in the GPU-less container: the native layers / optimizer / BaseModel wiring is
run end to end with the kernels replaced by the torch-CPU emulation in tests/_cpu_emul.py (test
infrastructure, monkeypatched in; the product has no such path) and must reproduce the golden
vectors recorded from the real reference.  This validates layout planning, autograd plumbing, the
exact-mode step protocol (begin_step -> de-dup -> catch-up -> gather -> ... -> clip -> updates ->
flush) — everything except the HIP kernels themselves, which tests/test_gpu_*.py cover."""
import numpy as np
import pytest
import torch

import _cpu_emul
from conftest import Golden, assert_weights_close, _din_fields


def _build(g, tmp_path, monkeypatch, sparse_update="exact"):
    _cpu_emul.install(monkeypatch)
    from fuxictr_amd import optim, zoo
    from fuxictr_amd.features import FeatureMap
    monkeypatch.setattr(optim._NativeOptimizer, "__init__", _cpu_opt_init(optim))
    m = g.meta
    fmap = FeatureMap(g.spec["dataset_id"], str(tmp_path))
    fmap.load_dict(g.spec, {"embedding_dim": m["embedding_dim"]})
    common = dict(gpu=-1, embedding_dim=m["embedding_dim"], learning_rate=m["lr"],
                  optimizer=m["optimizer"], loss="binary_crossentropy",
                  task="binary_classification", metrics=["logloss", "AUC"], verbose=0,
                  model_root=str(tmp_path), sparse_update=sparse_update,
                  embedding_regularizer=m.get("emb_reg", 0), net_regularizer=m.get("net_reg", 0))
    if m["model"] == "DeepFM":
        model = zoo.DeepFM(fmap, model_id=m["name"], hidden_units=m["hidden"],
                           batch_norm=m.get("batch_norm", False), **common)
    elif m["model"] == "xDeepFM":
        model = zoo.xDeepFM(fmap, model_id=m["name"], dnn_hidden_units=m["hidden"],
                            cin_hidden_units=m["cin"], **common)
    elif m["model"] == "DLRM":
        model = zoo.DLRM(fmap, model_id=m["name"], top_mlp_units=m["hidden"],
                         bottom_mlp_units=m["bottom"], interaction_op=m.get("interaction_op", "dot"), **common)
    elif m["model"] == "DIN":
        model = zoo.DIN(fmap, model_id=m["name"], dnn_hidden_units=m["hidden"],
                        dnn_activations="relu", attention_hidden_units=m["att_hidden"],
                        attention_hidden_activations="Dice", din_target_field=_din_fields(m, "din_target", "adgroup_id"),
                        din_sequence_field=_din_fields(m, "din_sequence", "click_sequence"),
                        din_use_softmax=m.get("din_softmax", False), **common)
    else:
        model = zoo.DCNv2(fmap, model_id=m["name"], model_structure=m.get("structure", "parallel"),
                          num_cross_layers=m["n_cross"], parallel_dnn_hidden_units=m["hidden"], stacked_dnn_hidden_units=m.get("stacked", []),
                          **common)
    sd = {k: torch.from_numpy(v) for k, v in g.state0.items()}
    assert sorted(model.state_dict().keys()) == sorted(sd.keys())
    model.load_state_dict(sd)
    model._max_gradient_norm = m["max_norm"]
    return model


def _cpu_opt_init(optim):
    """The product optimizer refuses CPU parameters; for the wiring tests lift exactly that check."""
    orig = optim._NativeOptimizer.__init__

    def init(self, params, lr, model=None, **kw):
        self._require_cuda = False
        orig(self, params, lr, model=model, **kw)
    return init


def tb(b):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in b.items()}


def test_wiring_forward(golden, tmp_path, monkeypatch):
    model = _build(golden, tmp_path, monkeypatch)
    model.eval()
    with torch.no_grad():
        p = model.forward(tb(golden.batches[-1]))["y_pred"]
    np.testing.assert_allclose(p._fx_logit.reshape(-1).numpy(), golden.expect["logit0"], atol=5e-6)


def test_wiring_training_trajectory(golden, tmp_path, monkeypatch):
    model = _build(golden, tmp_path, monkeypatch)
    model.train()
    losses = [float(model.train_step(tb(golden.batches[i])).item())
              for i in range(golden.meta["steps"])]
    np.testing.assert_allclose(losses, golden.expect["loss"], atol=5e-6)
    model.eval()
    with torch.no_grad():
        p = model.forward(tb(golden.batches[-1]))["y_pred"]
    np.testing.assert_allclose(p.reshape(-1).numpy(), golden.expect["pred1"], atol=1e-5)
    sd = model.state_dict()
    for k, ref in golden.state1.items():
        assert_weights_close(sd[k].numpy(), ref, golden.meta["lr"], golden.meta["steps"], k)


def test_pooled_sequences_are_reduced_inside_the_gather(tmp_path, monkeypatch):
    """a sequence feature behind MaskedSum/AveragePooling takes ONE slot of the
    gather record — the pooling launch runs, the torch encoders are never called and the record is
    handed to the model without a stack/cat pass."""
    g = Golden("deepfm_seqpool")
    model = _build(g, tmp_path, monkeypatch)
    import fuxictr_amd.layers as nat
    import fuxictr_amd.ops as ops
    calls = {"pool": 0, "dedup": 0}
    real_pool, real_dedup = ops.emb_seq_pool_fwd, ops.dedup

    def counting_pool(*a, **k):
        calls["pool"] += 1
        return real_pool(*a, **k)

    def counting_dedup(*a, **k):
        calls["dedup"] += 1
        return real_dedup(*a, **k)
    monkeypatch.setattr(ops, "emb_seq_pool_fwd", counting_pool)
    monkeypatch.setattr(ops, "dedup", counting_dedup)

    def never(self, *a, **k):
        raise AssertionError("torch pooling encoder called on the native path")
    monkeypatch.setattr(nat.MaskedAveragePooling, "forward", never)
    monkeypatch.setattr(nat.MaskedSumPooling, "forward", never)
    layer = model.embedding_layer.embedding_layer
    X = model.get_inputs(tb(g.batches[0]))
    model.train()
    d = layer(X)
    (rec, plan), = d._records
    n_feats = len(g.features)
    assert plan.n_slots == n_feats and rec.shape[1:] == (n_feats, g.meta["embedding_dim"])
    assert d["click_sequence"].shape == d["cate_sequence"].shape == d["userid"].shape
    assert layer.dict2tensor(d).data_ptr() == rec.data_ptr()      # the record itself, no copy
    model.train_step(tb(g.batches[0]))
    # per forward: 1 pooling launch for both sequences; per training forward one de-dup per table
    # group (the LR copy does not share click_sequence's table — use_sharing=False — so its keys
    # differ from the D=8 group's and it cannot reuse that de-dup here)
    assert calls["pool"] == 2 and calls["dedup"] == 3, calls


MIXED_SEQ_SPEC = {
    "dataset_id": "mix", "num_fields": 7, "total_features": 0, "input_length": 0, "labels": ["y"],
    "features": [
        {"first": {"source": "", "type": "sequence", "padding_idx": 0, "vocab_size": 14,
                   "max_len": 3, "feature_encoder": "layers.MaskedAveragePooling()"}},
        {"price": {"source": "", "type": "numeric"}},
        {"item": {"source": "", "type": "categorical", "padding_idx": 0, "vocab_size": 31}},
        {"hist_mean": {"source": "", "type": "sequence", "padding_idx": 0, "vocab_size": 31,
                       "max_len": 6, "share_embedding": "item",
                       "feature_encoder": "layers.MaskedAveragePooling()"}},
        {"hist_raw": {"source": "", "type": "sequence", "padding_idx": 0, "vocab_size": 31,
                      "max_len": 4, "share_embedding": "item", "feature_encoder": None}},
        {"one": {"source": "", "type": "sequence", "padding_idx": 0, "vocab_size": 9, "max_len": 1,
                 "feature_encoder": "layers.MaskedSumPooling()"}},
        {"user": {"source": "", "type": "categorical", "vocab_size": 12}},
        {"tags": {"source": "", "type": "sequence", "padding_idx": 0, "vocab_size": 17, "max_len": 9,
                  "feature_encoder": "layers.MaskedAveragePooling()"}},
    ]}


@pytest.mark.parametrize("D", [10, 8, 3])
def test_mixed_pooled_and_raw_sequences_forward_and_row_gradients(D, tmp_path, monkeypatch):
    """Plan logic of SURVEY.md 8f-3 at the layer level, on the kernel emulation: mean-/sum-pooled
    histories (one sharing the item table, one of length 1, one after the last plain column), a raw
    [B, L, D] history (DIN style) and plain fields in ONE table group — every entry of the returned
    dict and the reduced per-row gradients equal torch's embedding + pooling + autograd."""
    _cpu_emul.install(monkeypatch)
    import fuxictr_amd.layers as nat
    from fuxictr_amd.features import FeatureMap
    fmap = FeatureMap("mix", str(tmp_path))
    fmap.load_dict(MIXED_SEQ_SPEC, {"embedding_dim": D})
    layer = nat.FeatureEmbeddingDict(fmap, D, embedding_initializer="partial(nn.init.normal_, std=0.5)")
    (grp,) = layer.table_groups()
    grp.opt_kind = "sgd"                      # what the optimizer would set: de-dup the lookups
    gen = torch.Generator().manual_seed(D)
    B = 37
    X = {}
    for item in MIXED_SEQ_SPEC["features"]:
        (name, fs), = item.items()
        if fs["type"] == "numeric":
            X[name] = torch.rand(B, generator=gen)
        elif fs["type"] == "sequence":
            ids = torch.randint(1, fs["vocab_size"], (B, fs["max_len"]), generator=gen)
            keep = torch.randint(0, fs["max_len"] + 1, (B,), generator=gen)
            ids[torch.arange(fs["max_len"]).view(1, -1) >= keep.view(-1, 1)] = 0
            X[name] = ids
        else:
            X[name] = torch.randint(0, fs["vocab_size"], (B,), generator=gen)
    layer.train()
    out = layer(nat.FeatureDict(X))
    # torch reference on leaf copies of the same tables
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in layer.state_dict().items()}
    feats = {k: v for it in MIXED_SEQ_SPEC["features"] for k, v in it.items()}
    owner = {"hist_mean": "item", "hist_raw": "item"}
    ref = {}
    for name, fs in feats.items():
        w = sd["embedding_layers.%s.weight" % owner.get(name, name)]
        if fs["type"] == "numeric":
            ref[name] = X[name].view(-1, 1) * w.view(1, -1)
            continue
        e = torch.nn.functional.embedding(X[name], w, padding_idx=fs.get("padding_idx"))
        enc = fs.get("feature_encoder")
        if enc == "layers.MaskedAveragePooling()":
            e = e.sum(1) / ((e.sum(-1) != 0).float().sum(-1, keepdim=True) + 1e-12)
        elif enc == "layers.MaskedSumPooling()":
            e = e.sum(1)
        ref[name] = e
    assert list(out.keys()) == list(ref.keys())
    (rec, plan), = out._records
    assert plan.n_slots == 7 + 4 and set(plan.pooled) == {"first", "hist_mean", "one", "tags"}
    assert plan.slot["first"] == (0, 1) and plan.slot["hist_raw"] == (4, 4)     # slots: feature order
    assert [f for f, _ in plan.id_feats] == ["item", "hist_raw", "user",          # pooled ids last
                                             "first", "hist_mean", "one", "tags"]
    for name in ref:
        assert out[name].shape == ref[name].shape, name
        np.testing.assert_allclose(out[name].detach().numpy(), ref[name].detach().numpy(),
                                   atol=1e-6, err_msg=name)
    # backward: random cotangents on every entry
    cot = {n: torch.randn(ref[n].shape, generator=gen) for n in ref}
    sum((out[n] * cot[n]).sum() for n in ref).backward()
    sum((ref[n] * cot[n]).sum() for n in ref).backward()
    (pend,) = grp.pending
    nu = int(pend.dd.n_unique)
    rows = pend.dd.uniq_row[:nu].long()
    dense = torch.zeros(grp.total_rows, D)
    for name, (base, V, pidx) in grp.tables.items():
        g = sd["embedding_layers.%s.weight" % name].grad.clone()
        if pidx is not None:
            g[pidx].zero_()
        dense[base:base + V] = g
    np.testing.assert_allclose(pend.G[:nu].numpy(), dense[rows].numpy(), atol=2e-6)
    untouched = torch.ones(grp.total_rows, dtype=torch.bool)
    untouched[rows] = False
    assert float(dense[untouched].abs().max()) == 0.0
    np.testing.assert_allclose(grp.num_grad.reshape(-1).numpy(),
                               sd["embedding_layers.price.weight"].grad.reshape(-1).numpy(),
                               atol=2e-6)


def test_entrywise_reads_of_the_record_share_one_autograd_node(tmp_path, monkeypatch):
    """Reading the dict entry by entry (DIN style) — including leaving most entries unused — goes
    through ONE backward node that assembles the record's gradient (zeros for unused views)."""
    _cpu_emul.install(monkeypatch)
    import fuxictr_amd.layers as nat
    from fuxictr_amd.features import FeatureMap
    fmap = FeatureMap("mix", str(tmp_path))
    fmap.load_dict(MIXED_SEQ_SPEC, {"embedding_dim": 4})
    layer = nat.FeatureEmbeddingDict(fmap, 4, embedding_initializer="partial(nn.init.normal_, std=0.5)")
    (grp,) = layer.table_groups()
    grp.opt_kind = "sgd"
    gen = torch.Generator().manual_seed(5)
    B = 9
    X = {}
    for item in MIXED_SEQ_SPEC["features"]:
        (name, fs), = item.items()
        if fs["type"] == "numeric":
            X[name] = torch.rand(B, generator=gen)
        elif fs["type"] == "sequence":
            X[name] = torch.randint(1, fs["vocab_size"], (B, fs["max_len"]), generator=gen)
        else:
            X[name] = torch.randint(1, fs["vocab_size"], (B,), generator=gen)
    layer.train()
    out = layer(nat.FeatureDict(X))
    nodes = {out[n].grad_fn for n in out}
    assert len(nodes) == 1 and type(next(iter(nodes))).__name__ == "_SplitRecordFnBackward"
    calls = []
    real = grp.backward

    def spy(plan, ids, dense, dout, *a, **k):
        calls.append(dout.clone())
        return real(plan, ids, dense, dout, *a, **k)
    monkeypatch.setattr(grp, "backward", spy)
    w_raw = torch.randn(B, 4, 4, generator=gen)
    w_usr = torch.randn(B, 4, generator=gen)
    ((out["hist_raw"] * w_raw).sum() + (out["user"] * w_usr).sum() + out["user"].sum()).backward()
    (dout,), (rec, plan) = calls, out._records[0]
    expect = torch.zeros(B, plan.n_slots, 4)
    s, w = plan.slot["hist_raw"]
    expect[:, s:s + w] = w_raw
    expect[:, plan.slot["user"][0]] = w_usr + 1.0
    assert torch.equal(dout.view(B, plan.n_slots, 4), expect)


def test_optimizer_step_protocol_of_a_train_step_override(tmp_path, monkeypatch):
    """Both loop shapes of the reference drive the native optimizer correctly:
      b: optimizer.zero_grad(); forward; backward; optimizer.step()      (rank_model.py:307-323)
      c: forward; backward; optimizer.step(); optimizer.zero_grad()      (the LongCTR models' own
         train_step, model_zoo/LongCTR/DCNv2/DCNv2.py:209-217 — no zero_grad before the FIRST step)
    In both the step counter / bias corrections advance before the forward reads a row (the first
    training forward opens the step when zero_grad has not), step() picks up fit()'s
    max_gradient_norm, and an explicit optimizer.set_max_norm() is not overridden by the model's."""
    g = Golden("deepfm_adam_clip")
    a = _build(g, tmp_path, monkeypatch)
    b = _build(g, tmp_path, monkeypatch)
    c = _build(g, tmp_path, monkeypatch)
    for m_ in (a, b, c):
        m_.train()
    b._max_gradient_norm = g.meta["max_norm"]
    c._max_gradient_norm = g.meta["max_norm"]
    for i in range(g.meta["steps"]):
        batch = tb(g.batches[i])
        la = float(a.train_step(batch).item())
        b.optimizer.zero_grad()                       # the override's own loop
        out = b.forward(batch)
        loss = b.compute_loss(out, b.get_labels(batch))
        loss.backward()
        b.optimizer.step()
        assert abs(float(loss.item()) - la) <= 1e-6, (i, float(loss.item()), la)
        assert abs(la - float(g.expect["loss"][i])) <= 1e-4
        out = c.forward(batch)                        # LongCTR order: zero_grad comes last
        loss_c = c.compute_loss(out, c.get_labels(batch))
        loss_c.backward()
        c.optimizer.step()
        c.optimizer.zero_grad()
        assert abs(float(loss_c.item()) - la) <= 1e-6, (i, float(loss_c.item()), la)
    # (a's train_step evaluates head + loss + head backward in one pass, layers._HeadCtx; the loops written
    # out above take the separate kernels — two summation orders of the same gradients: the tolerance of
    # every weight comparison of this suite, conftest.assert_weights_close)
    sa, sc_ = a.state_dict(), c.state_dict()
    for k in sa:
        assert_weights_close(sa[k].float().numpy(), sc_[k].float().numpy(), g.meta["lr"], g.meta["steps"], k)
    # an lr change between two steps of the LongCTR order (zero_grad already opened the next step)
    # still reaches the step it precedes
    for m_ in (a, c):
        for grp in m_.optimizer.param_groups:
            grp["lr"] = grp["lr"] * 0.1
    batch = tb(g.batches[0])
    la = float(a.train_step(batch).item())
    out = c.forward(batch)
    loss_c = c.compute_loss(out, c.get_labels(batch))
    loss_c.backward()
    c.optimizer.step()
    c.optimizer.zero_grad()
    # (a missed lr change would move every touched element by ~0.9 lr: far outside the 2e-5 the bulk of a
    # tensor has to meet)
    sa, sc_ = a.state_dict(), c.state_dict()
    for k in sa:
        assert_weights_close(sa[k].float().numpy(), sc_[k].float().numpy(), g.meta["lr"],
                             g.meta["steps"] + 1, k)
    # explicit clip setting wins over the model attribute
    c.optimizer.set_max_norm(0.0)
    out = c.forward(batch)
    c.compute_loss(out, c.get_labels(batch)).backward()
    c.optimizer.step()
    assert c.optimizer._max_norm == 0.0


def test_bf16_table_storage_wiring(tmp_path, monkeypatch):
    """`emb_dtype: bf16` on the emulated kernels: D > 1 tables are bf16, the D=1 LR tables and the
    Adam state stay fp32, a reference (fp32) checkpoint loads into it, the forward equals the oracle
    on the widened weights, a few training steps run through the fused path."""
    from oracle import ctr_oracle as O
    g = Golden("deepfm_adam")
    _cpu_emul.install(monkeypatch)
    from fuxictr_amd import optim, zoo
    from fuxictr_amd.features import FeatureMap
    monkeypatch.setattr(optim._NativeOptimizer, "__init__", _cpu_opt_init(optim))
    m = g.meta
    fmap = FeatureMap(g.spec["dataset_id"], str(tmp_path))
    fmap.load_dict(g.spec, {"embedding_dim": m["embedding_dim"]})
    model = zoo.DeepFM(fmap, model_id="bf16", gpu=-1, embedding_dim=m["embedding_dim"],
                       hidden_units=m["hidden"], learning_rate=m["lr"], optimizer="adam",
                       loss="binary_crossentropy", task="binary_classification",
                       metrics=["logloss", "AUC"], verbose=0, model_root=str(tmp_path),
                       emb_dtype="bf16")
    model.load_state_dict({k: torch.from_numpy(v) for k, v in g.state0.items()})
    main = model.embedding_layer.embedding_layer.table_groups()[0]
    lr_grp = model.fm.lr_layer.embedding_layer.embedding_layer.table_groups()[0]
    assert main.table.dtype == torch.bfloat16 and lr_grp.table.dtype == torch.float32
    assert main.m.dtype == torch.float32
    wide = {k: v.detach().float().clone() for k, v in model.state_dict().items()}
    tr = O.OracleTrainer(g.cfg(), wide, g.features, lr=m["lr"], max_norm=m["max_norm"])
    b = tb(g.batches[-1])
    model.eval()
    with torch.no_grad():
        p = model.forward(b)["y_pred"]
    np.testing.assert_allclose(p._fx_logit.reshape(-1).numpy(), tr.logits(b).numpy(), atol=5e-6)
    model.train()
    model._max_gradient_norm = m["max_norm"]
    for i in range(3):
        loss = float(model.train_step(tb(g.batches[i])).item())
        ref = tr.train_step(tb(g.batches[i]), tb(g.batches[i])["label"])[0]
        assert abs(loss - ref) < 2e-2, (i, loss, ref)       # bf16 rounding of the rows, not exactness
    model.eval()       # flush through adam_catchup_all
    assert main.table.dtype == torch.bfloat16


def test_din_attention_runs_inside_the_gather_record(tmp_path, monkeypatch):
    """The reference's DIN configuration (one target, one raw sequence as the last feature) takes the
    in-record path: the tower reads a prefix of the record, its input gradient lands in the record's
    gradient buffer (no copy), and the trajectory equals the unfused composition's bit for bit on the
    CPU emulation (same arithmetic, other memory layout)."""
    from conftest import Golden
    from fuxictr_amd import layers as L
    g = Golden("din_adam")
    calls = {"fwd": 0, "inplace": 0}
    fwd0, bwd0 = L._DinRecordFn.forward, L._DinRecordFn.backward

    def fwd(ctx, *a):
        calls["fwd"] += 1
        return fwd0(ctx, *a)

    def bwd(ctx, dflat):
        buf = ctx.grad_slot.buf
        calls["inplace"] += int(buf is not None and dflat.data_ptr() == buf.data_ptr())
        return bwd0(ctx, dflat)
    monkeypatch.setattr(L._DinRecordFn, "forward", staticmethod(fwd))
    monkeypatch.setattr(L._DinRecordFn, "backward", staticmethod(bwd))
    model = _build(g, tmp_path, monkeypatch)
    assert model._in_record == ("adgroup_id", "click_sequence")
    model.train()
    losses = [float(model.train_step(tb(g.batches[i])).item()) for i in range(g.meta["steps"])]
    assert calls["fwd"] == g.meta["steps"] and calls["inplace"] == g.meta["steps"]
    np.testing.assert_allclose(losses, g.expect["loss"], atol=5e-6)
    sd = model.state_dict()
    monkeypatch.setenv("FX_DIN_INPLACE", "0")
    other = _build(g, tmp_path, monkeypatch)
    assert other._in_record is None
    other.train()
    losses0 = [float(other.train_step(tb(g.batches[i])).item()) for i in range(g.meta["steps"])]
    np.testing.assert_allclose(losses, losses0, atol=1e-6)
    for k, v in other.state_dict().items():
        np.testing.assert_allclose(sd[k].numpy(), v.numpy(), atol=2e-6, err_msg=k)
    assert int(sd["attention_layers.0.attention_layer.mlp.1.bn.num_batches_tracked"]) == g.meta["steps"]


def test_din_record_with_a_reserved_slot_also_serves_the_general_composition(tmp_path, monkeypatch):
    """The reserved slot changes the record's layout whether or not a forward takes the in-record path
    (e.g. an attention MLP the fused kernels do not cover): the general composition — views of the
    record, concatenation, per-view gradients — must give the same trajectory on that layout."""
    from conftest import Golden
    from fuxictr_amd import layers as L
    g = Golden("din_adam")
    monkeypatch.setattr(L.DIN_Attention, "forward_in_record", lambda self, *a, **k: None)
    model = _build(g, tmp_path, monkeypatch)
    assert model._in_record is not None                      # the slot IS reserved
    model.train()
    losses = [float(model.train_step(tb(g.batches[i])).item()) for i in range(g.meta["steps"])]
    np.testing.assert_allclose(losses, g.expect["loss"], atol=5e-6)
    model.eval()                    # (exact mode: leaving training mode settles the pending row replays)
    sd = {k: v.numpy() for k, v in model.state_dict().items()}
    for name, got, ref in g.final_weights(sd):
        assert_weights_close(got, ref, g.meta["lr"], g.meta["steps"], name)


@pytest.mark.parametrize("case", ["dlrm_adam", "dlrm_sparse_only", "dlrm_cat"])
def test_dlrm_bottom_vector_lives_in_the_gather_record(case, tmp_path, monkeypatch):
    """DLRM with numeric features and the dot interaction: the bottom tower writes its vector into the
    record's reserved last slot, one launch produces [dots | vector | padding] for the top tower and one
    its gradient.  Same trajectory as the concatenating composition (FX_DLRM_INPLACE=0); models without
    numeric features / with the `cat` interaction keep the general path."""
    from conftest import Golden
    from fuxictr_amd import layers as L
    g = Golden(case)
    calls = {"n": 0}
    fwd0 = L._DlrmMixFn.forward

    def fwd(ctx, *a):
        calls["n"] += 1
        return fwd0(ctx, *a)
    monkeypatch.setattr(L._DlrmMixFn, "forward", staticmethod(fwd))
    model = _build(g, tmp_path, monkeypatch)
    model.train()
    losses = [float(model.train_step(tb(g.batches[i])).item()) for i in range(g.meta["steps"])]
    np.testing.assert_allclose(losses, g.expect["loss"], atol=5e-6)
    expect_fast = case == "dlrm_adam"
    assert model._in_record == expect_fast and calls["n"] == (g.meta["steps"] if expect_fast else 0)
    model.eval()
    sd = {k: v.numpy() for k, v in model.state_dict().items()}
    for name, got, ref in g.final_weights(sd):
        assert_weights_close(got, ref, g.meta["lr"], g.meta["steps"], name)
    if expect_fast:
        monkeypatch.setenv("FX_DLRM_INPLACE", "0")
        other = _build(g, tmp_path, monkeypatch)
        other.train()
        losses0 = [float(other.train_step(tb(g.batches[i])).item()) for i in range(g.meta["steps"])]
        np.testing.assert_allclose(losses, losses0, atol=1e-6)


@pytest.mark.parametrize("case,fused_steps", [("deepfm_adam", 3), ("dcnv2_adam", 3), ("din_adam", 3),
                                              ("dlrm_adam", 3), ("xdeepfm_adam", 3)])
def test_training_head_takes_the_one_pass_kernel(case, fused_steps, tmp_path, monkeypatch):
    """layers._HeadCtx: in a train_step the tower that produces the logit runs head forward + BCE + head
    backward as ops.head_train (no ops.sigmoid_bce launch); a Linear(K -> 1) whose output is NOT the logit
    (xDeepFM's CIN `fc`: the DNN's head adds to it afterwards) takes the offer once, is found out in the
    backward pass and stops asking; evaluation never takes it."""
    import fuxictr_amd.ops as real
    g = Golden(case)
    model = _build(g, tmp_path, monkeypatch)
    calls = {"head": 0, "bce": 0}
    emul_head, emul_bce = real.head_train, real.sigmoid_bce

    def head(*a, **k):
        calls["head"] += 1
        return emul_head(*a, **k)

    def bce(*a, **k):
        if a[1] is not None:            # (y is None: activation only, evaluate / predict)
            calls["bce"] += 1
        return emul_bce(*a, **k)
    monkeypatch.setattr(real, "head_train", head)
    monkeypatch.setattr(real, "sigmoid_bce", bce)
    model.train()
    per_step = []
    for i in range(3):
        calls["head"] = 0
        model.train_step(tb(g.batches[i]))
        per_step.append(calls["head"])
    assert calls["bce"] == 0, calls
    assert per_step == [1, 1, 1], per_step
    calls["head"] = 0
    model.eval()
    with torch.no_grad():
        model.forward(tb(g.batches[0]))
    assert calls["head"] == 0


def test_a_head_whose_output_is_not_the_logit_falls_back(monkeypatch):
    """A Linear(K -> 1) that takes the step's offer (layers._HeadCtx) although more arithmetic follows its
    output (xDeepFM's CIN `fc` when a DNN head adds to it, the reference DeepFM's `y_pred += mlp(...)`): the
    gradient that reaches it is not the fused dlogit, so its backward takes the separate kernels — same
    gradients as without the offer — and the module stops asking."""
    _cpu_emul.install(monkeypatch)
    from fuxictr_amd import layers
    gen = torch.Generator().manual_seed(5)
    B, K = 12, 8
    mlp = layers.MLP_Block(input_dim=K, output_dim=1, hidden_units=[16], hidden_activations="ReLU")
    x = torch.randn(B, K, generator=gen, requires_grad=True)
    y = (torch.rand(B, 1, generator=gen) > 0.5).float()

    def run(offer):
        for p_ in mlp.parameters():
            p_.grad = None
        x.grad = None
        layers._HEAD_CTX = layers._HeadCtx(lambda: y, 1.0, 0) if offer else None
        try:
            out = mlp(x)
        finally:
            layers._HEAD_CTX = None
        (out * 2.0 + 1.0).sigmoid().sum().backward()
        return [x.grad.clone()] + [p_.grad.clone() for p_ in mlp.parameters()]
    ref = run(False)
    got = run(True)
    assert mlp.__dict__.get("_fx_head_off") is True
    for a_, b_ in zip(got, ref):
        assert torch.equal(a_, b_)
    calls = []
    monkeypatch.setattr(layers.ops, "head_train", lambda *a, **k: calls.append(1))
    run(True)
    assert not calls


def test_a_logit_modified_in_place_after_the_head_is_not_the_fused_logit(monkeypatch):
    """the reference's AutoInt / DESTINE write `y_pred = self.fc(...); y_pred += self.lr_layer(X)`
    (AutoInt.py:112-116, DESTINE.py:133-135).  The in-place add leaves the logit's storage (and `data_ptr()`) where
    the fused head put its result, with another value in it: the loss must not take the head's loss / dlogit
    (formed on the value before the add), the tower's backward must not take the head's gradients.  The tensor's
    version counter tells the two apart: same loss and same gradients as without the offer."""
    _cpu_emul.install(monkeypatch)
    from fuxictr_amd import layers, rank_model
    gen = torch.Generator().manual_seed(9)
    B, K = 16, 8
    fc = layers.MLP_Block(input_dim=K, output_dim=1, hidden_units=[12], hidden_activations="ReLU")
    wide = torch.nn.Linear(K, 1)                       # a stock layer: nothing overwrites the head's result
    act = rank_model.FxSigmoid()
    x = torch.randn(B, K, generator=gen)
    y = (torch.rand(B, 1, generator=gen) > 0.5).float()

    def run(offer):
        for p_ in list(fc.parameters()) + list(wide.parameters()):
            p_.grad = None
        layers._HEAD_CTX = layers._HeadCtx(lambda: y, 1.0, 0) if offer else None
        try:
            logit = fc(x)
            logit += wide(x)                           # IN PLACE, after the head ran
            loss = rank_model._bce_loss(act(logit), y)
        finally:
            layers._HEAD_CTX = None
        loss.backward()
        return [loss.detach().clone()] + [p_.grad.clone() for p_ in list(fc.parameters()) + list(wide.parameters())]
    ref = run(False)
    got = run(True)
    for a_, b_ in zip(got, ref):
        assert torch.allclose(a_, b_, rtol=1e-6, atol=1e-7), (a_, b_)
    want = torch.nn.functional.binary_cross_entropy(torch.sigmoid(fc(x) + wide(x)).detach(), y)
    assert abs(float(got[0]) - float(want)) <= 1e-6


def test_dcnv2_head_masks_the_deep_columns_itself(tmp_path, monkeypatch):
    """zoo.DCNv2 `parallel`: the head reads the [cross | deep] buffer of layers._CrossDeepFn; the one-pass
    head (ops.head_train, mask_from = width of the cross part) hands back a gradient whose deep columns
    already carry the top ReLU's mask, so _CrossDeepFn.backward launches no ops.mask_mul of its own
    (layers._ReluNote).  With FX_HEAD_FUSED off the launch is there."""
    import fuxictr_amd.ops as real
    from fuxictr_amd import layers
    g = Golden("dcnv2_adam")
    counts = {}
    for fused in (True, False):
        monkeypatch.setattr(layers, "_HEAD_FUSED", fused)
        model = _build(g, tmp_path, monkeypatch)
        calls = {"mask": 0, "from": []}
        emul_mask, emul_head = real.mask_mul, real.head_train

        def mask(*a, **k):
            calls["mask"] += 1
            return emul_mask(*a, **k)

        def head(*a, **k):
            calls["from"].append(a[5])
            return emul_head(*a, **k)
        monkeypatch.setattr(real, "mask_mul", mask)
        monkeypatch.setattr(real, "head_train", head)
        model.train()
        losses = [float(model.train_step(tb(g.batches[i])).item()) for i in range(g.meta["steps"])]
        np.testing.assert_allclose(losses, g.expect["loss"], rtol=0, atol=1e-4)
        counts[fused] = (calls["mask"], list(calls["from"]))
        monkeypatch.setattr(real, "mask_mul", emul_mask)
        monkeypatch.setattr(real, "head_train", emul_head)
    assert counts[True][0] == 0 and all(f > 0 for f in counts[True][1]), counts
    assert counts[False][0] == g.meta["steps"] and not counts[False][1], counts


def test_relu_note_is_ignored_when_the_buffer_has_a_second_consumer(monkeypatch):
    """layers._ReluNote: the [cross | deep] node skips its own ReLU-mask launch only when the gradient it
    receives is the very tensor the fused head produced.  With a second consumer of the buffer autograd hands it
    the SUM of two gradients (a new tensor): the mask launch runs and the gradients equal the unfused ones."""
    _cpu_emul.install(monkeypatch)
    from fuxictr_amd import layers
    import fuxictr_amd.rank_model as rm
    import fuxictr_amd.ops as real
    gen = torch.Generator().manual_seed(11)
    B, D0, H = 16, 8, 12
    x0 = torch.randn(B, D0, generator=gen, requires_grad=True)
    Wc, bc = torch.randn(D0, D0, generator=gen, requires_grad=True), torch.randn(D0, generator=gen, requires_grad=True)
    Wd, bd = torch.randn(H, D0, generator=gen, requires_grad=True), torch.randn(H, generator=gen, requires_grad=True)
    fc = layers.FxLinear(D0 + H, 1)
    y = (torch.rand(B, 1, generator=gen) > 0.5).float()
    c = torch.randn(B, D0 + H, generator=gen)
    leaves = [x0, Wc, bc, Wd, bd, fc.weight, fc.bias]
    unit = rm._unit_grad(torch.device("cpu"))
    masks = []
    emul_mask = real.mask_mul
    monkeypatch.setattr(real, "mask_mul", lambda *a, **k: (masks.append(1), emul_mask(*a, **k))[1])

    def run(offer, second_consumer):
        for t in leaves:
            t.grad = None
        del masks[:]
        layers._HEAD_CTX = layers._HeadCtx(lambda: y, 1.0, unit.data_ptr()) if offer else None
        try:
            out = layers._CrossDeepFn.apply(x0, 1, (True,), Wc, bc, Wd, bd)
            logit = fc(out)
            pred = logit.view_as(logit)
            pred._fx_logit = logit
            loss = rm._bce_loss(pred, y)
        finally:
            layers._HEAD_CTX = None
            layers._RELU_NOTES.clear()
        if second_consumer:
            loss = loss + (out * c).sum()
        loss.backward(gradient=unit)        # (the root gradient the step announced; an add hands it on as it is)
        return [t.grad.clone() for t in leaves], len(masks)
    ref2, n_ref2 = run(False, True)
    got2, n_got2 = run(True, True)
    assert n_ref2 == 1 and n_got2 == 1                 # the node masked the summed gradient itself
    for a_, b_ in zip(got2, ref2):
        assert torch.allclose(a_, b_, atol=1e-6, rtol=1e-6)
    ref1, n_ref1 = run(False, False)
    got1, n_got1 = run(True, False)
    assert n_ref1 == 1 and n_got1 == 0                 # sole consumer: the head's gradient is read in place
    for a_, b_ in zip(got1, ref1):
        assert torch.allclose(a_, b_, atol=1e-6, rtol=1e-6)


def test_row_record_host_logic(tmp_path, monkeypatch):
    """Round 6's row record on the emulated kernels: with an exact-mode Adam attached, table / m / v / last_step
    of every fp32 table group are column ranges of ONE [rows, W] record, the per-feature Parameters are views of
    it, state_dict keys / values are the reference's, save_weights writes compact tables, a dtype change of the
    module goes back to four packed arrays — and the training trajectory still is the golden one (the layout
    moves no bit)."""
    import os
    from fuxictr_amd import optim
    from fuxictr_amd.layers import _TableGroup
    g = Golden("deepfm_adam")
    assert [_TableGroup.record_width(d) for d in (1, 4, 8, 10, 16, 32, 64)] == [4, 16, 28, 32, 64, 128, 224]
    model = _build(g, tmp_path, monkeypatch)
    groups = [grp for grp in model.optimizer._groups if grp.table is not None]
    assert groups and all(grp.record is not None for grp in groups)
    for grp in groups:
        W = _TableGroup.record_width(grp.D)
        assert grp.record.shape == (grp.table.shape[0], W)
        assert grp.table.data_ptr() == grp.record.data_ptr() and grp.table.stride(0) == W
        assert grp.m.stride(0) == W and grp.v.stride(0) == W
        assert grp.last_step.dtype == torch.int32 and grp.last_step.stride(0) == W
        assert float(grp.m.abs().sum()) == 0.0 and int(grp.last_step.abs().sum()) == 0
    # the Parameters alias the record: writing one shows in the group's table
    sd = model.state_dict()
    key = [k for k in sd if ".embedding_layers." in k and sd[k].dim() == 2 and sd[k].shape[1] > 1][0]
    before = sd[key][1].clone()
    sd[key][1] += 1.0
    grp16 = [grp for grp in groups if grp.D > 1][0]
    assert any(torch.equal(grp16.table[r], before + 1.0) for r in range(grp16.table.shape[0]))
    sd[key][1] -= 1.0
    # the golden trajectory on the record layout, then the same without it
    model.train()
    losses = [float(model.train_step(tb(g.batches[i])).item()) for i in range(g.meta["steps"])]
    np.testing.assert_allclose(losses, g.expect["loss"], atol=2e-5)
    monkeypatch.setattr(optim, "ROW_RECORD", False)
    packed = _build(g, tmp_path / "packed", monkeypatch)
    assert all(grp.record is None for grp in packed.optimizer._groups)
    packed.train()
    losses_p = [float(packed.train_step(tb(g.batches[i])).item()) for i in range(g.meta["steps"])]
    # (ATen's CPU kernels — the emulation — are not layout-invariant to the last bit: strided and packed
    # operands take different vector paths; the HIP kernels are, tests/test_gpu_row_record.py holds them to
    # torch.equal)
    np.testing.assert_allclose(losses, losses_p, rtol=0, atol=1e-6)
    for mdl in (model, packed):               # both layouts end at the reference's weights
        mdl.optimizer.flush()                 # (exact mode: rows a late batch did not touch are caught up now)
        sd1 = mdl.state_dict()
        for k, ref in g.state1.items():
            assert_weights_close(sd1[k].numpy(), ref, g.meta["lr"], g.meta["steps"], k)
    # compact checkpoints
    pa, pb = str(tmp_path / "a" / "m.model"), str(tmp_path / "b" / "m.model")
    model.save_weights(pa)
    packed.save_weights(pb)
    assert abs(os.path.getsize(pa) - os.path.getsize(pb)) < 65536
    # a dtype change leaves the record layout (it is fp32 by construction) and keeps the values
    ref = {k: v.clone() for k, v in model.state_dict().items()}
    model.double()
    assert all(grp.record is None for grp in groups)
    for k, v in model.state_dict().items():
        assert v.dtype == torch.float64 or not v.is_floating_point()
        assert torch.equal(v.float(), ref[k].float()), k
