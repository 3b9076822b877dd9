"""Generate golden vectors by running the REAL reference (reczoo/FuxiCTR at /root/reference).

Run in the build container only (the reference does not travel to the GPU box):
    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python3 -B /root/repo/tests/golden/make_golden.py
It imports fuxictr + model_zoo from /root/reference (with empty stub modules for polars / h5py /
keras_preprocessing, none of which is touched on this path — SURVEY.md §8c), builds the reference's
own DeepFM / DCNv2 on CPU, runs forward + `train_step` (rank_model.py:307-323) on small seeded
synthetic batches and stores inputs, initial weights and the reference's outputs in
tests/golden/<case>.npz.  Nothing is written inside /root/reference.
"""
import json
import os
import sys
import types

# FX_GOLDEN_OUT=<dir>: regenerate somewhere else (tests/golden/check_regen.py compares with the
# committed fixtures)
OUT_DIR = os.environ.get("FX_GOLDEN_OUT") or os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
TMP = "/tmp/fx_golden"


def _import_reference():
    for name in ["polars", "h5py", "keras_preprocessing", "keras_preprocessing.sequence"]:
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["keras_preprocessing.sequence"].pad_sequences = lambda *a, **k: None
    sys.modules["keras_preprocessing"].sequence = sys.modules["keras_preprocessing.sequence"]
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)


def small_criteo_spec(dataset_id, n_dense, cards):
    feats = []
    for j in range(n_dense):
        feats.append({"I%d" % (j + 1): {"source": "", "type": "numeric"}})
    for c, card in enumerate(cards):
        feats.append({"C%d" % (c + 1): {"source": "", "type": "categorical", "padding_idx": 0,
                                       "vocab_size": int(card) + 1}})
    return {"dataset_id": dataset_id, "num_fields": len(feats), "total_features": 0,
            "input_length": len(feats), "labels": ["label"], "features": feats}


def small_seq_spec(dataset_id, L=6):
    """tiny_seq-like schema: a sequence feature sharing the item table, padding_idx 0."""
    feats = [
        {"price": {"source": "item", "type": "numeric"}},
        {"userid": {"source": "user", "type": "categorical", "padding_idx": 0, "vocab_size": 301}},
        {"adgroup_id": {"source": "item", "type": "categorical", "padding_idx": 0, "vocab_size": 97}},
        {"cate_id": {"source": "item", "type": "categorical", "padding_idx": 0, "vocab_size": 23}},
        {"pid": {"source": "context", "type": "categorical", "padding_idx": 0, "vocab_size": 4}},
        {"click_sequence": {"source": "user", "type": "sequence", "feature_encoder": None,
                            "share_embedding": "adgroup_id", "padding_idx": 0, "vocab_size": 97,
                            "max_len": L}},
    ]
    return {"dataset_id": dataset_id, "num_fields": len(feats), "total_features": 0,
            "input_length": 0, "labels": ["label"], "features": feats}


def pooled_seq_spec(dataset_id, L=7):
    """Sequence features behind the pooling encoders (the default for every sequence feature,
    feature_processor.py:379): a mean-pooled history sharing the item table, a sum-pooled one with
    its own table and max_len, all padded with 0 (some histories empty)."""
    feats = [
        {"price": {"source": "item", "type": "numeric"}},
        {"userid": {"source": "user", "type": "categorical", "padding_idx": 0, "vocab_size": 301}},
        {"adgroup_id": {"source": "item", "type": "categorical", "padding_idx": 0, "vocab_size": 97}},
        {"click_sequence": {"source": "user", "type": "sequence",
                            "feature_encoder": "layers.MaskedAveragePooling()",
                            "share_embedding": "adgroup_id", "padding_idx": 0, "vocab_size": 97,
                            "max_len": L}},
        {"cate_id": {"source": "item", "type": "categorical", "padding_idx": 0, "vocab_size": 23}},
        {"cate_sequence": {"source": "user", "type": "sequence",
                           "feature_encoder": "layers.MaskedSumPooling()", "padding_idx": 0,
                           "vocab_size": 41, "max_len": L + 4}},
        {"pid": {"source": "context", "type": "categorical", "padding_idx": 0, "vocab_size": 4}},
    ]
    return {"dataset_id": dataset_id, "num_fields": len(feats), "total_features": 0,
            "input_length": 0, "labels": ["label"], "features": feats}


def pair_seq_spec(dataset_id, L=5):
    """Two aligned histories (item ids and their categories) for DIN's grouped fields:
    din_target_field (adgroup_id, cate_id) against din_sequence_field (click_sequence, cate_sequence)."""
    spec = small_seq_spec(dataset_id, L=L)
    spec["features"].append({"cate_sequence": {"source": "user", "type": "sequence",
                                               "feature_encoder": None, "share_embedding": "cate_id",
                                               "padding_idx": 0, "vocab_size": 23, "max_len": L}})
    spec["num_fields"] = len(spec["features"])
    return spec


def make_batches(rng, spec, B, n, pad_frac=0.02):
    import numpy as np
    batches = []
    for _ in range(n):
        b = {}
        for item in spec["features"]:
            (name, fs), = item.items()
            if fs["type"] == "numeric":
                b[name] = rng.random(B, dtype=np.float32)
            elif fs["type"] == "sequence":
                card, L = fs["vocab_size"] - 1, fs["max_len"]
                ids = rng.integers(1, card + 1, size=(B, L)).astype(np.int64)
                lens = rng.integers(0, L + 1, size=B)          # post-padded with 0, some empty
                ids[np.arange(L)[None, :] >= lens[:, None]] = 0
                b[name] = ids
            else:
                card = fs["vocab_size"] - 1
                ids = np.floor(card * rng.random(B) ** 3).astype(np.int64) + 1
                ids = np.minimum(ids, card)
                ids[rng.random(B) < pad_frac] = 0     # padding_idx occurrences
                b[name] = ids
        b["label"] = (rng.random(B) < 0.3).astype(np.float32)
        batches.append(b)
    return batches


def run_case(case):
    import numpy as np
    import torch
    from fuxictr.features import FeatureMap
    from fuxictr.pytorch.torch_utils import seed_everything
    name = case["name"]
    if case["model"] == "DIN" and case.get("schema") == "pair_seq":
        spec = pair_seq_spec(name)
    elif case["model"] == "DIN":
        spec = small_seq_spec(name)
    elif case.get("schema") == "pooled_seq":
        spec = pooled_seq_spec(name)
    else:
        spec = small_criteo_spec(name, case["n_dense"], case["cards"])
    for item in spec["features"]:          # per-feature embedding_dim (feature_embedding.py:140)
        (fname, fs), = item.items()
        if fname in case.get("feature_dims", {}):
            fs["embedding_dim"] = case["feature_dims"][fname]
    os.makedirs(os.path.join(TMP, name), exist_ok=True)
    fm_path = os.path.join(TMP, name, "feature_map.json")
    with open(fm_path, "w") as f:
        json.dump(spec, f)
    seed_everything(case["seed"])
    torch.set_num_threads(8)
    fmap = FeatureMap(name, os.path.join(TMP, name))
    fmap.load(fm_path, {"embedding_dim": case["embedding_dim"]})
    common = dict(gpu=-1, embedding_dim=case["embedding_dim"], learning_rate=case["lr"],
                  optimizer=case["optimizer"], loss="binary_crossentropy",
                  task="binary_classification", metrics=["logloss", "AUC"], verbose=0,
                  model_root=TMP, embedding_regularizer=case.get("emb_reg", 0),
                  net_regularizer=case.get("net_reg", 0))
    if case["model"] == "DeepFM":
        from model_zoo.DeepFM.DeepFM_torch.src import DeepFM
        model = DeepFM(fmap, model_id=name, hidden_units=case["hidden"],
                       batch_norm=case.get("batch_norm", False), **common)
    elif case["model"] == "xDeepFM":
        from model_zoo import xDeepFM
        model = xDeepFM(fmap, model_id=name, dnn_hidden_units=case["hidden"],
                        cin_hidden_units=case["cin"], **common)
    elif case["model"] == "DLRM":
        from model_zoo import DLRM
        model = DLRM(fmap, model_id=name, top_mlp_units=case["hidden"],
                     bottom_mlp_units=case["bottom"], interaction_op=case.get("interaction_op", "dot"), **common)
    elif case["model"] == "DIN":
        from model_zoo import DIN
        model = DIN(fmap, model_id=name, dnn_hidden_units=case["hidden"], dnn_activations="relu",
                    attention_hidden_units=case["att_hidden"],
                    attention_hidden_activations="Dice",
                    din_target_field=[tuple(f) if isinstance(f, list) else f
                                      for f in case.get("din_target", ["adgroup_id"])],
                    din_sequence_field=[tuple(f) if isinstance(f, list) else f
                                        for f in case.get("din_sequence", ["click_sequence"])],
                    din_use_softmax=case.get("din_softmax", False), **common)
    else:
        from model_zoo import DCNv2
        model = DCNv2(fmap, model_id=name, model_structure=case.get("structure", "parallel"),
                      num_cross_layers=case["n_cross"],
                      parallel_dnn_hidden_units=case["hidden"],
                      stacked_dnn_hidden_units=case.get("stacked", []), **common)
    # make the (1e-4 std) tables matter numerically: rescale so logits are O(1)
    with torch.no_grad():
        for k, p in model.named_parameters():
            if "embedding_layers" in k and "lr_layer" not in k and p.shape[0] > 1 \
                    and p.dim() == 2 and p.shape[1] > 1:
                p.mul_(case.get("emb_scale", 1.0))
            # (the LR copy of a `share_embedding` feature keeps nn.Embedding's N(0,1) init — the
            # reference's init_weights skips it, feature_embedding.py:207-209 — leave that one)
            if "lr_layer" in k and "embedding_layers" in k and p.shape[0] > 1 \
                    and float(p.abs().mean()) < 1e-2:
                p.mul_(case.get("lr_scale", 1.0))
    model._max_gradient_norm = case["max_norm"]          # what fit() would set (rank_model.py:251)
    logits = []
    model.output_activation.register_forward_pre_hook(lambda m, inp: logits.append(inp[0].detach().clone()))
    if case["model"] == "DIN":       # make the Dice gate and its alpha non-trivial
        with torch.no_grad():
            for k, p in model.named_parameters():
                if k.endswith(".alpha"):
                    p.uniform_(-0.5, 0.5)
    rng = np.random.default_rng(case["seed"])
    batches = make_batches(rng, spec, case["B"], case["steps"] + 1)
    out = {}
    for k, v in model.state_dict().items():
        out["state0/" + k] = v.detach().cpu().numpy().copy()
    def to_torch(b):
        out_b = {k: torch.from_numpy(v) for k, v in b.items()}
        if case["model"] == "DLRM":      # DLRM.py:114 concatenates X[k] on dim -1: needs [B,1]
            for item in spec["features"]:
                (fname, fs), = item.items()
                if fs["type"] == "numeric":
                    out_b[fname] = out_b[fname].view(-1, 1)
        return out_b
    model.eval()
    with torch.no_grad():
        p0 = model.forward(to_torch(batches[-1]))["y_pred"]
    out["expect/pred0"] = p0.numpy().reshape(-1).copy()
    out["expect/logit0"] = logits[-1].numpy().reshape(-1).copy()
    model.train()
    losses, norms = [], []
    for i in range(case["steps"]):
        b = to_torch(batches[i])
        # total grad norm of this step, measured the way clip_grad_norm_ does
        loss = model.train_step(b)
        losses.append(float(loss.item()))
    out["expect/loss"] = np.asarray(losses, dtype=np.float64)
    model.eval()
    with torch.no_grad():
        p1 = model.forward(to_torch(batches[-1]))["y_pred"]
    out["expect/pred1"] = p1.numpy().reshape(-1).copy()
    out["expect/logit1"] = logits[-1].numpy().reshape(-1).copy()
    for k, v in model.state_dict().items():
        out["state1/" + k] = v.detach().cpu().numpy().copy()
    for i, b in enumerate(batches):
        for k, v in b.items():
            out["batch%d/%s" % (i, k)] = v
    meta = dict(case)
    meta["spec"] = spec
    meta["torch"] = torch.__version__
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(OUT_DIR, name + ".npz")
    np.savez_compressed(path, **out)
    print(name, "loss", losses, "pred0[:3]", out["expect/pred0"][:3], "pred1[:3]",
          out["expect/pred1"][:3], "->", path, os.path.getsize(path) // 1024, "KiB")


SLIM_ROW_STRIDE = 8

DEMOS = {
    # BASELINE.json configs[0]: demo/example3_DeepFM_with_npz_input.py, its own config and data
    "c1_tiny_npz": dict(module="model_zoo.DeepFM.DeepFM_torch.src", cls="DeepFM",
                        config="demo/config/example3_config", expid="DeepFM_test_npz",
                        data="tiny_npz"),
    # the model zoo's own smoke configs (model_zoo/<M>/config, `<M>_test`).  DIN_test runs on its own
    # data/tiny_seq (npz).  The other three name data/tiny_parquet, whose loader needs polars (absent
    # here): same model config, data swapped to data/tiny_npz.
    "demo_din_tiny_seq": dict(module="model_zoo", cls="DIN", config="model_zoo/DIN/config",
                              expid="DIN_test", data="tiny_seq"),
    "demo_dcnv2_tiny_npz": dict(module="model_zoo", cls="DCNv2", config="model_zoo/DCNv2/config",
                                expid="DCNv2_test", data="tiny_npz", swap_data=True, slim=True),
    "demo_xdeepfm_tiny_npz": dict(module="model_zoo", cls="xDeepFM",
                                  config="model_zoo/xDeepFM/config", expid="xDeepFM_test",
                                  data="tiny_npz", swap_data=True),
    "demo_dlrm_tiny_npz": dict(module="model_zoo", cls="DLRM", config="model_zoo/DLRM/config",
                               expid="DLRM_test", data="tiny_npz", swap_data=True),
    # the training-control loop of BaseModel.fit (rank_model.py:236-306) over several epochs: the
    # c1 configuration with a learning rate large enough for the validation AUC to stall, so that
    # reduce-lr-on-plateau, the best-only checkpoint, early stopping and the final reload of the
    # best weights all happen (monitor = AUC - logloss keeps the decisions away from AUC's ties)
    "fit_control_tiny_npz": dict(module="model_zoo.DeepFM.DeepFM_torch.src", cls="DeepFM",
                                 config="demo/config/example3_config", expid="DeepFM_test_npz",
                                 data="tiny_npz",
                                 overrides=dict(epochs=8, learning_rate=0.02, early_stop_patience=3,
                                                monitor={"AUC": 1, "logloss": -1})),
}


def _demo_meta(name, cls, params, n_steps):
    """The per-model hyper-parameters under the key names tests/conftest.py:Golden.cfg reads."""
    meta = dict(name=name, model=cls, embedding_dim=params["embedding_dim"],
                lr=params["learning_rate"], optimizer=params["optimizer"], max_norm=10.0,
                steps=n_steps, B=params["batch_size"], seed=params["seed"],
                emb_reg=params["embedding_regularizer"], net_reg=params["net_regularizer"],
                expid=params.get("model_id"))
    if cls == "DeepFM":
        meta["hidden"] = list(params["hidden_units"])
    elif cls == "DIN":
        meta.update(hidden=list(params["dnn_hidden_units"]),
                    att_hidden=list(params["attention_hidden_units"]))
    elif cls == "DCNv2":
        meta.update(hidden=list(params["parallel_dnn_hidden_units"]),
                    n_cross=params["num_cross_layers"])
    elif cls == "xDeepFM":
        # (xDeepFM_test says `cin_layer_units`, which the model ignores: it runs the default CIN)
        meta.update(hidden=list(params["dnn_hidden_units"]),
                    cin=list(params.get("cin_hidden_units", [16, 16, 16])))
    elif cls == "DLRM":
        meta.update(hidden=list(params["top_mlp_units"]), bottom=list(params["bottom_mlp_units"]))
    return meta


def run_demo(name):
    """One of the reference's own example / smoke configurations run through its own
    RankDataLoader + BaseModel.fit + evaluate.  Stored: the batches fit() actually saw (in shuffled
    order), the validation set, initial/final weights and the reference's logloss/AUC."""
    import importlib
    import numpy as np
    import torch
    from fuxictr.features import FeatureMap
    from fuxictr.pytorch.dataloaders import RankDataLoader
    from fuxictr.pytorch.torch_utils import seed_everything
    from fuxictr.utils import load_config
    d = DEMOS[name]
    Model = getattr(importlib.import_module(d["module"]), d["cls"])
    data_dir = os.path.join(REF, "data", d["data"])
    if d.get("swap_data"):
        import yaml
        with open(os.path.join(REF, d["config"], "model_config.yaml")) as f:
            params = yaml.safe_load(f)[d["expid"]]
        params.update(model_id=d["expid"], dataset_id=d["data"], data_format="npz",
                      data_root=os.path.join(REF, "data"))
        for k in ("train", "valid", "test"):
            params[k + "_data"] = os.path.join(data_dir, k + ".npz")
    else:
        params = load_config(os.path.join(REF, d["config"]), d["expid"])
        for k in ("train_data", "valid_data", "test_data"):
            params[k] = os.path.join(data_dir, os.path.basename(params[k]))
    params.update(d.get("overrides", {}))
    params["gpu"] = -1
    params["model_root"] = os.path.join(TMP, name)
    os.makedirs(os.path.join(TMP, name, params["dataset_id"]), exist_ok=True)
    params["num_workers"] = 0
    params["verbose"] = 0
    fmap = FeatureMap(params["dataset_id"], data_dir)
    fmap.load(os.path.join(data_dir, "feature_map.json"), params)
    with open(os.path.join(data_dir, "feature_map.json")) as f:
        spec = json.load(f)
    seed_everything(params["seed"])
    model = Model(fmap, **params)
    out = {}
    for k, v in model.state_dict().items():
        out["state0/" + k] = v.detach().cpu().numpy().copy()
    train_gen, valid_gen = RankDataLoader(fmap, stage="train", **params).make_iterator()
    seen, losses = [], []
    inner = model.train_step

    def recording_step(batch_data):
        seen.append({k: v.detach().clone() for k, v in batch_data.items()})
        loss = inner(batch_data)
        losses.append(float(loss.item()))
        return loss
    model.train_step = recording_step
    evals, lrs = [], []
    inner_eval = model.evaluate

    def recording_eval(gen, metrics=None):
        logs = inner_eval(gen, metrics=metrics)
        # fit() evaluates with the MONITOR's metrics only (rank_model.py:301): a run monitored on
        # AUC alone has no logloss here
        evals.append([float(logs.get("logloss", np.nan)), float(logs.get("AUC", np.nan))])
        return logs
    inner_ckpt = model.checkpoint_and_earlystop

    def recording_ckpt(logs, **kw):
        inner_ckpt(logs, **kw)
        lrs.append(float(model.optimizer.param_groups[0]["lr"]))
    model.evaluate = recording_eval
    model.checkpoint_and_earlystop = recording_ckpt
    model.fit(train_gen, validation_data=valid_gen, **params)
    model.train_step = inner
    model.evaluate = inner_eval
    model.checkpoint_and_earlystop = inner_ckpt
    if evals and not np.isnan(np.asarray(evals)).any():
        # the training-control record (fit_control_*): every in-fit evaluation + the lr after it
        out["expect/fit_evals"] = np.asarray(evals, dtype=np.float64)
        out["expect/fit_lrs"] = np.asarray(lrs, dtype=np.float64)
    res = model.evaluate(valid_gen)
    valid = {}
    for b in valid_gen:
        for k, v in b.items():
            valid.setdefault(k, []).append(v)
    valid = {k: torch.cat(v) for k, v in valid.items()}
    model.eval()
    with torch.no_grad():
        p1 = model.forward(valid)["y_pred"]
    out["expect/pred1"] = p1.numpy().reshape(-1).copy()
    out["expect/loss"] = np.asarray(losses, dtype=np.float64)
    out["expect/valid_logloss"] = np.asarray([res["logloss"]], dtype=np.float64)
    out["expect/valid_auc"] = np.asarray([res["AUC"]], dtype=np.float64)
    for k, v in model.state_dict().items():
        v = v.detach().cpu().numpy().copy()
        if d.get("slim") and v.ndim == 2 and v.size > 65536:
            # the trained copy of a big matrix: every 8th row (state0 holds the full random init —
            # incompressible — and the update of a matrix is checked as well on a row sample; keeps
            # the fixture small enough to be committed)
            out["state1s/" + k] = v[::SLIM_ROW_STRIDE].copy()
        else:
            out["state1/" + k] = v
    for i, b in enumerate(seen + [valid]):
        for k, v in b.items():
            out["batch%d/%s" % (i, k)] = v.numpy()
    meta = _demo_meta(name, d["cls"], params, len(seen))
    meta.update(epochs=params["epochs"], early_stop_patience=params.get("early_stop_patience", 2),
                monitor=params["monitor"], steps_per_epoch=len(train_gen))
    meta["spec"] = spec
    meta["torch"] = torch.__version__
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(OUT_DIR, name + ".npz")
    np.savez_compressed(path, **out)
    print(name, "steps", len(seen), "loss", losses, "valid", dict(res), "->", path,
          os.path.getsize(path) // 1024, "KiB")


CARDS = [37, 13, 1500, 900, 11, 5, 211, 19, 3, 401, 97, 1200, 53, 7]
CASES = [
    dict(name="deepfm_adam", model="DeepFM", n_dense=5, cards=CARDS, embedding_dim=8,
         hidden=[64, 32], B=192, steps=6, lr=1e-2, optimizer="adam", max_norm=10.0, seed=2019,
         emb_scale=1000.0, lr_scale=1000.0),
    dict(name="deepfm_adam_clip", model="DeepFM", n_dense=5, cards=CARDS, embedding_dim=8,
         hidden=[64, 32], B=192, steps=4, lr=1e-2, optimizer="adam", max_norm=0.05, seed=7,
         emb_scale=1000.0, lr_scale=1000.0),
    dict(name="deepfm_sgd", model="DeepFM", n_dense=5, cards=CARDS, embedding_dim=8,
         hidden=[64, 32], B=192, steps=4, lr=5e-2, optimizer="SGD", max_norm=10.0, seed=11,
         emb_scale=1000.0, lr_scale=1000.0),
    dict(name="deepfm_d10", model="DeepFM", n_dense=3, cards=CARDS[:9], embedding_dim=10,
         hidden=[48], B=100, steps=3, lr=1e-2, optimizer="adam", max_norm=10.0, seed=3,
         emb_scale=1000.0, lr_scale=1000.0),
    dict(name="deepfm_reg", model="DeepFM", n_dense=3, cards=CARDS[:8], embedding_dim=8,
         hidden=[32, 16], B=128, steps=5, lr=1e-2, optimizer="adam", max_norm=10.0, seed=23,
         emb_scale=1000.0, lr_scale=1000.0, emb_reg=1e-3, net_reg="l1_l2(1e-5,1e-4)"),
    dict(name="deepfm_reg_sgd", model="DeepFM", n_dense=3, cards=CARDS[:8], embedding_dim=8,
         hidden=[32, 16], B=128, steps=5, lr=5e-2, optimizer="SGD", max_norm=0.05, seed=29,
         emb_scale=1000.0, lr_scale=1000.0, emb_reg="l1_l2(1e-4,1e-2)", net_reg=0),
    # BatchNorm case trains with SGD: the bias of a Linear that feeds BatchNorm has an exactly-zero
    # gradient (BN subtracts the batch mean), autograd delivers fp32 summation residue (~1e-9), and
    # Adam would normalise that residue into +-lr steps whose sign depends on the BLAS build — the
    # reference's own value of those biases (and of running_mean, which absorbs them) is noise.
    dict(name="deepfm_bn", model="DeepFM", n_dense=3, cards=CARDS[:8], embedding_dim=8,
         hidden=[32, 16], B=128, steps=4, lr=5e-2, optimizer="SGD", max_norm=10.0, seed=31,
         emb_scale=1000.0, lr_scale=1000.0, batch_norm=True),
    dict(name="deepfm_seqpool", model="DeepFM", schema="pooled_seq", embedding_dim=8,
         hidden=[32, 16], B=160, steps=4, lr=1e-2, optimizer="adam", max_norm=10.0, seed=37,
         emb_scale=1000.0, lr_scale=1000.0),
    dict(name="xdeepfm_adam", model="xDeepFM", n_dense=3, cards=CARDS[:9], embedding_dim=8,
         hidden=[32, 16], cin=[12, 6, 5], B=128, steps=4, lr=1e-2, optimizer="adam", max_norm=10.0,
         seed=17, emb_scale=1000.0, lr_scale=1000.0),
    dict(name="dlrm_adam", model="DLRM", n_dense=5, cards=CARDS, embedding_dim=8,
         hidden=[64, 32], bottom=[32, 16], B=192, steps=5, lr=1e-2, optimizer="adam",
         max_norm=10.0, seed=13, emb_scale=1000.0),
    # DLRM's other interaction ('cat' = flatten) and a schema without numeric features (no bottom MLP)
    dict(name="dlrm_cat", model="DLRM", interaction_op="cat", n_dense=3, cards=CARDS[:6],
         embedding_dim=4, hidden=[16, 8], bottom=[8], B=80, steps=4, lr=1e-2, optimizer="adam",
         max_norm=10.0, seed=59, emb_scale=1000.0),
    dict(name="dlrm_sparse_only", model="DLRM", n_dense=0, cards=CARDS[:7], embedding_dim=4,
         hidden=[16, 8], bottom=[8], B=80, steps=4, lr=1e-2, optimizer="adam", max_norm=10.0,
         seed=61, emb_scale=1000.0),
    dict(name="din_adam", model="DIN", embedding_dim=8, hidden=[32, 16], att_hidden=[16], B=160,
         steps=5, lr=1e-2, optimizer="adam", max_norm=10.0, seed=5, emb_scale=1000.0),
    # per-feature embedding dims: three table groups (D = 8, 4, 12; one numeric feature at D = 4
    # too), DCNv2 concatenates them (flatten_emb) to a 5*8 + ... wide input
    dict(name="dcnv2_mixdim", model="DCNv2", n_dense=3, cards=CARDS[:8], embedding_dim=8,
         hidden=[32, 16], n_cross=2, B=96, steps=4, lr=1e-2, optimizer="adam", max_norm=10.0,
         seed=41, emb_scale=1000.0, feature_dims={"C2": 4, "C6": 4, "I2": 4, "C4": 12}),
    # the other model_structure values of DCNv2.py:78-103: cross -> stacked DNN next to a parallel DNN
    # (narrow towers keep the fixture small), and the cross network alone
    dict(name="dcnv2_stacked_parallel", model="DCNv2", structure="stacked_parallel", n_dense=3,
         cards=CARDS[:6], embedding_dim=4, hidden=[16, 8], stacked=[12, 6], n_cross=2, B=80,
         steps=4, lr=1e-2, optimizer="adam", max_norm=10.0, seed=43, emb_scale=1000.0),
    dict(name="dcnv2_crossnet_only", model="DCNv2", structure="crossnet_only", n_dense=3,
         cards=CARDS[:6], embedding_dim=4, hidden=[], n_cross=3, B=80, steps=4, lr=1e-2,
         optimizer="adam", max_norm=10.0, seed=47, emb_scale=1000.0),
    # grouped DIN fields (item, category) attended together, softmax attention weights
    dict(name="din_pairs_softmax", model="DIN", schema="pair_seq", embedding_dim=4, hidden=[16, 8],
         att_hidden=[8], B=96, steps=4, lr=1e-2, optimizer="adam", max_norm=10.0, seed=53,
         emb_scale=1000.0, din_target=[["adgroup_id", "cate_id"]],
         din_sequence=[["click_sequence", "cate_sequence"]], din_softmax=True),
    dict(name="dcnv2_adam", model="DCNv2", n_dense=5, cards=CARDS, embedding_dim=8,
         hidden=[64, 32], n_cross=3, B=192, steps=5, lr=1e-2, optimizer="adam", max_norm=10.0,
         seed=2019, emb_scale=1000.0),
]

if __name__ == "__main__":
    _import_reference()
    only = sys.argv[1:]
    for case in CASES:
        if not only or case["name"] in only:
            run_case(case)
    for name in DEMOS:
        if not only or name in only:
            run_demo(name)
