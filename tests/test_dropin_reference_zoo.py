"""Drop-in check against the reference's OWN model classes (build container only: skipped where
/root/reference is absent, e.g. on the GPU box).  After `fuxictr_amd.patch.install()` the
reference's unmodified `model_zoo` DeepFM / DCNv2 are constructed from the native layers and — with
the kernels replaced by the test-only CPU emulation — reproduce the golden vectors that the same
classes produced on the stock torch layers."""
import os
import sys
import types

import numpy as np
import pytest
import torch

import _cpu_emul
from conftest import Golden, assert_weights_close, _din_fields
from test_host_wiring import _cpu_opt_init, tb

REF = os.environ.get("FX_REFERENCE_ROOT", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "fuxictr")),
                                reason="reference checkout not present")


def _fresh_reference(monkeypatch):
    for name in ["polars", "h5py", "keras_preprocessing", "keras_preprocessing.sequence"]:
        if name not in sys.modules:
            monkeypatch.setitem(sys.modules, name, types.ModuleType(name))
    sys.modules["keras_preprocessing.sequence"].pad_sequences = lambda *a, **k: None
    monkeypatch.setattr(sys, "dont_write_bytecode", True)
    monkeypatch.syspath_prepend(REF)
    # a fresh import of the reference under the patch
    for k in [k for k in sys.modules if k.startswith(("fuxictr.", "model_zoo")) or k == "fuxictr"]:
        monkeypatch.delitem(sys.modules, k)


def _drop_reference():
    for k in [k for k in sys.modules if k.startswith(("fuxictr.", "model_zoo")) or k == "fuxictr"]:
        sys.modules.pop(k, None)


@pytest.fixture
def patched_reference(monkeypatch):
    _fresh_reference(monkeypatch)
    _cpu_emul.install(monkeypatch)
    from fuxictr_amd import optim, patch
    monkeypatch.setattr(optim._NativeOptimizer, "__init__", _cpu_opt_init(optim))
    patch.install()
    yield
    _drop_reference()


@pytest.fixture
def patched_reference_gpu(monkeypatch):
    """The same patch with the REAL kernels: needs a GPU and a reference checkout on the same machine
    (the driver's GPU box has no /root/reference, so this variant runs wherever both exist)."""
    _fresh_reference(monkeypatch)
    from fuxictr_amd import patch
    patch.install()
    yield
    _drop_reference()


@pytest.mark.parametrize("case", ["deepfm_adam", "dcnv2_adam", "din_adam", "dlrm_adam",
                                  "xdeepfm_adam", "deepfm_reg", "deepfm_bn", "deepfm_seqpool", "dcnv2_mixdim", "dcnv2_stacked_parallel",
                                  "dcnv2_crossnet_only", "din_pairs_softmax", "dlrm_cat",
                                  "dlrm_sparse_only"])
def test_reference_zoo_classes_run_on_native_layers(case, patched_reference, tmp_path):
    _zoo_case(case, tmp_path, gpu=-1)


@pytest.mark.gpu
@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU next to the reference checkout")
@pytest.mark.parametrize("case", ["deepfm_adam", "dcnv2_adam", "din_adam", "dlrm_adam",
                                  "xdeepfm_adam", "deepfm_seqpool", "dcnv2_mixdim"])
def test_reference_zoo_classes_run_on_the_hip_kernels(case, patched_reference_gpu, tmp_path):
    """`patch.install()` + the reference's own model_zoo classes on the real kernels."""
    _zoo_case(case, tmp_path, gpu=0)


def _zoo_case(case, tmp_path, gpu):
    g = Golden(case)
    m = g.meta
    from fuxictr.features import FeatureMap
    import fuxictr_amd.layers as nat
    fmap = FeatureMap(g.spec["dataset_id"], str(tmp_path))
    fmap.load_dict(g.spec, {"embedding_dim": m["embedding_dim"]})
    common = dict(gpu=gpu, embedding_dim=m["embedding_dim"], learning_rate=m["lr"],
                  optimizer=m["optimizer"], loss="binary_crossentropy",
                  task="binary_classification", metrics=["logloss", "AUC"], verbose=0,
                  model_root=str(tmp_path), embedding_regularizer=m.get("emb_reg", 0),
                  net_regularizer=m.get("net_reg", 0))
    tol = 1.0 if gpu < 0 else 20.0      # GPU: different fp32 summation orders (tests/test_gpu_models.py)
    if m["model"] == "DeepFM":
        from model_zoo.DeepFM.DeepFM_torch.src import DeepFM as RefModel
        model = RefModel(fmap, model_id=case, hidden_units=m["hidden"],
                         batch_norm=m.get("batch_norm", False), **common)
    elif m["model"] == "DIN":
        from model_zoo import DIN as RefModel
        model = RefModel(fmap, model_id=case, dnn_hidden_units=m["hidden"], dnn_activations="relu",
                         attention_hidden_units=m["att_hidden"],
                         attention_hidden_activations="Dice", din_target_field=_din_fields(m, "din_target", "adgroup_id"),
                         din_sequence_field=_din_fields(m, "din_sequence", "click_sequence"),
                        din_use_softmax=m.get("din_softmax", False), **common)
    elif m["model"] == "DLRM":
        from model_zoo import DLRM as RefModel
        model = RefModel(fmap, model_id=case, top_mlp_units=m["hidden"],
                         bottom_mlp_units=m["bottom"], interaction_op=m.get("interaction_op", "dot"), **common)
    elif m["model"] == "xDeepFM":
        from model_zoo import xDeepFM as RefModel
        model = RefModel(fmap, model_id=case, dnn_hidden_units=m["hidden"],
                         cin_hidden_units=m["cin"], **common)
    else:
        from model_zoo import DCNv2 as RefModel
        model = RefModel(fmap, model_id=case, model_structure=m.get("structure", "parallel"),
                         num_cross_layers=m["n_cross"], parallel_dnn_hidden_units=m["hidden"], stacked_dnn_hidden_units=m.get("stacked", []),
                         **common)
    assert RefModel.__module__.startswith("model_zoo")           # the reference's own class ...
    assert isinstance(model.embedding_layer, (nat.FeatureEmbedding, nat.FeatureEmbeddingDict))  # native
    model.load_state_dict({k: torch.from_numpy(v) for k, v in g.state0.items()})
    model._max_gradient_norm = m["max_norm"]
    model.eval()
    def batch(i):
        b = tb(g.batches[i])
        if m["model"] == "DLRM":          # DLRM.py:114 concatenates X[k] on dim -1: needs [B,1]
            for name, spec in g.features.items():
                if spec["type"] == "numeric":
                    b[name] = b[name].view(-1, 1)
        return b
    with torch.no_grad():
        p = model.forward(batch(-1))["y_pred"]
    np.testing.assert_allclose(p.reshape(-1).cpu().numpy(), g.expect["pred0"], atol=2e-6 * tol)
    model.train()
    losses = [float(model.train_step(batch(i)).item()) for i in range(m["steps"])]
    np.testing.assert_allclose(losses, g.expect["loss"], atol=5e-6 * tol)
    model.eval()
    sd = model.state_dict()
    for k, ref in g.state1.items():
        assert_weights_close(sd[k].cpu().numpy(), ref, m["lr"], m["steps"], k)


def test_device_loader_through_the_reference_rankdataloader_hook(patched_reference):
    """`RankDataLoader(..., data_loader=DeviceNpzDataLoader)` (rank_dataloader.py:51-52) on the
    reference's own data/tiny_npz: same samples as its NpzDataLoader, column by column."""
    from fuxictr.features import FeatureMap
    from fuxictr.pytorch.dataloaders import RankDataLoader
    from fuxictr_amd.dataloader import DeviceNpzDataLoader
    data_dir = os.path.join(REF, "data", "tiny_npz")
    fmap = FeatureMap("tiny_npz", data_dir)
    fmap.load(os.path.join(data_dir, "feature_map.json"), {})
    paths = dict(train_data=os.path.join(data_dir, "train.npz"),
                 valid_data=os.path.join(data_dir, "valid.npz"))
    ours_t, ours_v = RankDataLoader(fmap, stage="train", batch_size=32, shuffle=False,
                                    data_loader=DeviceNpzDataLoader, device="cpu",
                                    **paths).make_iterator()
    ref_t, ref_v = RankDataLoader(fmap, stage="train", batch_size=32, shuffle=False, num_workers=0,
                                  **paths).make_iterator()
    for ours, ref in ((ours_t, ref_t), (ours_v, ref_v)):
        assert len(ours) == len(ref) and ours.num_samples == ref.num_samples
        for a, b in zip(ours, ref):
            assert set(a) == set(b)
            for k in b:
                np.testing.assert_array_equal(a[k].numpy().astype(np.float64), b[k].numpy())


def test_run_expid_flow_on_native_layers(patched_reference, tmp_path):
    """The body of the reference's run_expid.py (model_zoo/DeepFM/DeepFM_torch/run_expid.py:47-80:
    load_config -> FeatureMap -> `src.<model>` -> RankDataLoader -> fit -> evaluate) with
    `fuxictr_amd.patch.install()` in front of it — INTEGRATION.md section 1 — on the reference's own
    demo config and data.  Starting from the weights its stock run started from, it ends on that
    run's validation metrics (SURVEY.md 8c known answer: logloss 0.6798385031, AUC 0.9661458333)."""
    from fuxictr.features import FeatureMap
    from fuxictr.pytorch.dataloaders import RankDataLoader
    from fuxictr.pytorch.torch_utils import seed_everything
    from fuxictr.utils import load_config
    from model_zoo.DeepFM.DeepFM_torch import src
    import fuxictr_amd.layers as nat
    g = Golden("c1_tiny_npz")
    params = load_config(os.path.join(REF, "demo/config/example3_config"), "DeepFM_test_npz")
    params.update(gpu=-1, model_root=str(tmp_path), num_workers=0, verbose=0)
    for k in ("train_data", "valid_data", "test_data"):
        params[k] = os.path.join(REF, "data/tiny_npz", os.path.basename(params[k]))
    seed_everything(seed=params["seed"])
    data_dir = os.path.join(REF, "data", params["dataset_id"])
    feature_map = FeatureMap(params["dataset_id"], data_dir)
    feature_map.load(os.path.join(data_dir, "feature_map.json"), params)
    model = getattr(src, params["model"])(feature_map, **params)
    assert type(model).__module__.startswith("model_zoo")
    assert isinstance(model.embedding_layer, nat.FeatureEmbedding)
    model.count_parameters()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in g.state0.items()})
    train_gen, valid_gen = RankDataLoader(feature_map, stage="train", **params).make_iterator()
    model.fit(train_gen, validation_data=valid_gen, **params)
    valid_result = model.evaluate(valid_gen)
    assert abs(valid_result["logloss"] - 0.6798385031) <= 1e-6
    assert round(valid_result["AUC"], 4) == 0.9661
    test_gen = RankDataLoader(feature_map, stage="test", **params).make_iterator()
    assert set(model.evaluate(test_gen)) == {"logloss", "AUC"}


@pytest.fixture
def stock_reference(monkeypatch):
    """The reference importable as it is (no patch): its own torch layers, next to the native ones."""
    for name in ["polars", "h5py", "keras_preprocessing", "keras_preprocessing.sequence"]:
        if name not in sys.modules:
            monkeypatch.setitem(sys.modules, name, types.ModuleType(name))
    sys.modules["keras_preprocessing.sequence"].pad_sequences = lambda *a, **k: None
    monkeypatch.setattr(sys, "dont_write_bytecode", True)
    monkeypatch.syspath_prepend(REF)
    for k in [k for k in sys.modules if k.startswith(("fuxictr.", "model_zoo")) or k == "fuxictr"]:
        monkeypatch.delitem(sys.modules, k)
    _cpu_emul.install(monkeypatch)
    yield
    for k in [k for k in sys.modules if k.startswith(("fuxictr.", "model_zoo")) or k == "fuxictr"]:
        sys.modules.pop(k, None)


FILTER_SPEC = {
    "dataset_id": "flt", "num_fields": 8, "total_features": 0, "input_length": 0, "labels": ["y"],
    "features": [
        {"u_id": {"source": "user", "type": "categorical", "padding_idx": 0, "vocab_size": 30}},
        {"u_age": {"source": "user", "type": "numeric"}},
        {"i_id": {"source": "item", "type": "categorical", "padding_idx": 0, "vocab_size": 50}},
        {"i_price": {"source": "item", "type": "numeric"}},
        {"u_hist": {"source": "user", "type": "sequence", "padding_idx": 0, "vocab_size": 50,
                    "max_len": 4, "share_embedding": "i_id",
                    "feature_encoder": "layers.MaskedAveragePooling()"}},
        {"c_hour": {"source": "context", "type": "categorical", "vocab_size": 24}},
        {"i_cat": {"source": "item", "type": "categorical", "padding_idx": 0, "vocab_size": 9}},
        {"qid": {"type": "meta"}},
    ]}


@pytest.mark.parametrize("flatten", [False, True])
@pytest.mark.parametrize("kw", [
    {}, {"feature_source": "user"}, {"feature_source": ["item", "context"]},
    {"feature_type": "categorical"}, {"feature_type": ["numeric", "sequence"]},
    {"feature_source": "item", "feature_type": "categorical"}, {"feature_source": "context"}])
def test_feature_filters_match_the_reference_layer(kw, flatten, stock_reference, tmp_path):
    """FeatureEmbedding.forward(X, feature_source, feature_type, flatten_emb) and
    FeatureEmbeddingDict.dict2tensor(feature_list=...) (feature_embedding.py:73-88, :230-259, :261-297)
    against the reference's own stock layer holding the same weights."""
    from fuxictr.features import FeatureMap as RefFeatureMap
    from fuxictr.pytorch.layers import FeatureEmbedding as RefEmbedding
    import fuxictr_amd.layers as nat
    from fuxictr_amd.features import FeatureMap
    D = 8
    rmap = RefFeatureMap("flt", str(tmp_path))
    p = tmp_path / "feature_map.json"
    import json
    p.write_text(json.dumps(FILTER_SPEC))
    rmap.load(str(p), {"embedding_dim": D})
    nmap = FeatureMap("flt", str(tmp_path))
    nmap.load_dict(FILTER_SPEC, {"embedding_dim": D})
    torch.manual_seed(3)
    ref = RefEmbedding(rmap, D, embedding_initializer="partial(nn.init.normal_, std=0.5)")
    ours = nat.FeatureEmbedding(nmap, D)
    assert sorted(ours.state_dict().keys()) == sorted(ref.state_dict().keys())
    ours.load_state_dict(ref.state_dict())
    gen = torch.Generator().manual_seed(11)
    B = 23
    X = {"u_id": torch.randint(0, 30, (B,), generator=gen),
         "u_age": torch.rand(B, generator=gen),
         "i_id": torch.randint(0, 50, (B,), generator=gen),
         "i_price": torch.rand(B, generator=gen),
         "u_hist": torch.randint(0, 50, (B, 4), generator=gen),
         "c_hour": torch.randint(0, 24, (B,), generator=gen),
         "i_cat": torch.randint(0, 9, (B,), generator=gen)}
    ref.eval()
    ours.eval()
    with torch.no_grad():
        want = ref(dict(X), flatten_emb=flatten, **kw)
        got = ours(nat.FeatureDict(X), flatten_emb=flatten, **kw)
        assert got.shape == want.shape
        np.testing.assert_allclose(got.numpy(), want.numpy(), atol=1e-6)
        # dict2tensor with an explicit feature_list on the full dict
        names = ["i_id", "u_hist", "c_hour"]
        want2 = ref.embedding_layer.dict2tensor(ref.embedding_layer(dict(X)), flatten_emb=flatten,
                                                feature_list=names)
        got2 = ours.embedding_layer.dict2tensor(ours.embedding_layer(nat.FeatureDict(X)),
                                                flatten_emb=flatten, feature_list=names)
        np.testing.assert_allclose(got2.numpy(), want2.numpy(), atol=1e-6)


def _pretrain_case(tmp_path, usage, freeze):
    """A schema with one `pretrained_emb` feature + a sequence sharing its table, and the files
    the reference's loader reads (feature_vocab.json, an .npz of key/value pairs)."""
    import json
    rng = np.random.default_rng(11)
    vocab = {"user": {"__PAD__": 0, "u1": 1},      # (the reference reads the key type off entry 1)
             "item": {"__PAD__": 0, **{"i%d" % j: j for j in range(1, 39)}, "__OOV__": 39}}
    with open(os.path.join(str(tmp_path), "feature_vocab.json"), "w") as fd:
        json.dump(vocab, fd)
    np.savez(os.path.join(str(tmp_path), "item_emb.npz"),
             key=np.array(["i%d" % j for j in range(1, 30)]),
             value=rng.normal(size=(29, 6)).astype(np.float32))
    item = {"source": "item", "type": "categorical", "padding_idx": 0, "vocab_size": 40,
            "oov_idx": 39, "freeze_emb": freeze, "pretrained_emb": "item_emb.npz",
            "pretrain_dim": 6, "pretrain_usage": usage}
    spec = {"dataset_id": "pre", "labels": ["label"], "features": [
        {"price": {"source": "item", "type": "numeric"}},
        {"user": {"source": "user", "type": "categorical", "padding_idx": 0, "vocab_size": 30}},
        {"item": item},
        {"hist": {"source": "user", "type": "sequence", "share_embedding": "item",
                  "feature_encoder": "layers.MaskedAveragePooling()", "padding_idx": 0,
                  "vocab_size": 40, "max_len": 5}},
        {"ctx": {"source": "context", "type": "categorical", "vocab_size": 7}}]}
    B = 64
    X = {"price": rng.random(B).astype(np.float32), "user": rng.integers(0, 30, B),
         "item": rng.integers(0, 40, B), "hist": rng.integers(0, 40, (B, 5)),
         "ctx": rng.integers(0, 7, B)}
    return spec, X


@pytest.mark.parametrize("usage,freeze", [("init", False), ("sum", False), ("concat", True)])
def test_pretrained_emb_is_delegated_to_the_stock_module(usage, freeze, monkeypatch, tmp_path):
    """a `pretrained_emb` feature (and a sequence sharing its table) is served
    by the reference's own PretrainedEmbedding inside the native FeatureEmbedding — same
    state_dict keys, same forward values and same gradients as the stock FeatureEmbedding
    (feature_embedding.py:156-171, pretrained_embedding.py:30-189), while the other features stay
    on the native gather."""
    for name in ["polars", "h5py", "keras_preprocessing", "keras_preprocessing.sequence"]:
        if name not in sys.modules:
            monkeypatch.setitem(sys.modules, name, types.ModuleType(name))
    sys.modules["keras_preprocessing.sequence"].pad_sequences = lambda *a, **k: None
    monkeypatch.setattr(sys, "dont_write_bytecode", True)
    monkeypatch.syspath_prepend(REF)
    for k in [k for k in sys.modules if k.startswith(("fuxictr.", "model_zoo")) or k == "fuxictr"]:
        monkeypatch.delitem(sys.modules, k)
    try:
        from fuxictr.pytorch.layers.embeddings.feature_embedding import FeatureEmbedding as StockFE
        from fuxictr.pytorch.layers.embeddings.pretrained_embedding import PretrainedEmbedding
        import fuxictr_amd.layers as nat
        from fuxictr_amd.features import FeatureMap
        _cpu_emul.install(monkeypatch)
        spec, X = _pretrain_case(tmp_path, usage, freeze)
        fmap = FeatureMap("pre", str(tmp_path))
        fmap.load_dict(spec, {"embedding_dim": 8})
        torch.manual_seed(3)
        stock = StockFE(fmap, 8)
        native = nat.FeatureEmbedding(fmap, 8)
        layer = native.embedding_layer
        assert type(layer.embedding_layers["item"]) is PretrainedEmbedding
        assert layer.embedding_layers["hist"] is layer.embedding_layers["item"]
        assert "item" not in layer._feat_group and "user" in layer._feat_group
        sd = stock.state_dict()
        assert sorted(native.state_dict().keys()) == sorted(sd.keys())
        native.load_state_dict(sd)
        Xt = tb(X)
        w = torch.randn(64, 5, 8, generator=torch.Generator().manual_seed(5))
        outs = []
        for mod in (stock, native):
            mod.train()
            mod.zero_grad()
            out = mod(Xt)
            assert out.shape == (64, 5, 8)
            (out * w).sum().backward()
            outs.append(out.detach())
        np.testing.assert_allclose(outs[1].numpy(), outs[0].numpy(), atol=1e-6)
        stock_params = dict(stock.named_parameters())
        checked = 0
        for name, p in native.named_parameters():
            if ".item." not in name or not p.requires_grad:
                continue
            np.testing.assert_allclose(p.grad.numpy(), stock_params[name].grad.numpy(), atol=1e-6)
            checked += 1
        assert checked >= (1 if usage == "init" else 2) - (1 if freeze else 0)
    finally:
        for k in [k for k in sys.modules if k.startswith(("fuxictr.", "model_zoo")) or k == "fuxictr"]:
            sys.modules.pop(k, None)


def test_pretrained_emb_without_the_reference_package_says_so(monkeypatch, tmp_path):
    import fuxictr_amd.layers as nat
    from fuxictr_amd.features import FeatureMap
    for k in [k for k in sys.modules if k.startswith("fuxictr.") or k == "fuxictr"]:
        monkeypatch.delitem(sys.modules, k)
    monkeypatch.setattr(sys, "path", [p for p in sys.path if os.path.abspath(p) != REF])
    spec, _ = _pretrain_case(tmp_path, "init", False)
    fmap = FeatureMap("pre", str(tmp_path))
    fmap.load_dict(spec, {"embedding_dim": 8})
    with pytest.raises(NotImplementedError, match="PretrainedEmbedding"):
        nat.FeatureEmbedding(fmap, 8)


@pytest.mark.parametrize("usage,freeze", [("sum", False), ("init", True)])
def test_model_with_a_delegated_pretrained_feature_trains(usage, freeze, monkeypatch, tmp_path):
    """The delegated module's parameters are ordinary dense parameters of the native optimizer
    (frozen ones are left alone), next to the native sparse-row tables of the other features."""
    for name in ["polars", "h5py", "keras_preprocessing", "keras_preprocessing.sequence"]:
        if name not in sys.modules:
            monkeypatch.setitem(sys.modules, name, types.ModuleType(name))
    sys.modules["keras_preprocessing.sequence"].pad_sequences = lambda *a, **k: None
    monkeypatch.setattr(sys, "dont_write_bytecode", True)
    monkeypatch.syspath_prepend(REF)
    for k in [k for k in sys.modules if k.startswith(("fuxictr.", "model_zoo")) or k == "fuxictr"]:
        monkeypatch.delitem(sys.modules, k)
    try:
        from fuxictr_amd import optim, zoo
        from fuxictr_amd.features import FeatureMap
        _cpu_emul.install(monkeypatch)
        monkeypatch.setattr(optim._NativeOptimizer, "__init__", _cpu_opt_init(optim))
        spec, X = _pretrain_case(tmp_path, usage, freeze)
        fmap = FeatureMap("pre", str(tmp_path))
        fmap.load_dict(spec, {"embedding_dim": 8})
        model = zoo.DeepFM(fmap, model_id="pre", gpu=-1, embedding_dim=8, hidden_units=[16, 8],
                           learning_rate=1e-2, optimizer="adam", loss="binary_crossentropy",
                           task="binary_classification", metrics=["logloss", "AUC"], verbose=0,
                           model_root=str(tmp_path))
        rng = np.random.default_rng(0)
        X["label"] = (rng.random(64) < 0.3).astype(np.float32)
        pre = model.embedding_layer.embedding_layer.embedding_layers["item"]
        w0 = pre.pretrain_embedding.weight.detach().clone()
        model.train()
        losses = [float(model.train_step(tb(X)).item()) for _ in range(8)]
        assert np.all(np.isfinite(losses)) and losses[-1] < losses[0]
        moved = float((pre.pretrain_embedding.weight.detach() - w0).abs().max())
        assert (moved == 0.0) if freeze else (moved > 0.0)
    finally:
        for k in [k for k in sys.modules if k.startswith(("fuxictr.", "model_zoo")) or k == "fuxictr"]:
            sys.modules.pop(k, None)
