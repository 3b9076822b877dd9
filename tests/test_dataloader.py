"""DeviceNpzDataLoader: same samples as the npz file, column dtypes the model wants,
every sample exactly once per epoch; on the GPU the batches are device-resident views."""
import numpy as np
import pytest
import torch

from conftest import Golden


def _write_npz(tmp_path, g, n):
    from make_golden import make_batches
    rng = np.random.default_rng(5)
    b = make_batches(rng, g.spec, n, 1)[0]
    path = str(tmp_path / "train.npz")
    np.savez(path, **b)
    return path, b


def _fmap(g, tmp_path):
    from fuxictr_amd.features import FeatureMap
    fmap = FeatureMap(g.spec["dataset_id"], str(tmp_path))
    fmap.load_dict(g.spec, {"embedding_dim": g.meta["embedding_dim"]})
    return fmap


@pytest.mark.parametrize("case", ["deepfm_adam", "din_adam"])
def test_loader_yields_every_sample_once_cpu(case, tmp_path):
    from fuxictr_amd.dataloader import DeviceNpzDataLoader
    g = Golden(case)
    path, full = _write_npz(tmp_path, g, 1000)
    fmap = _fmap(g, tmp_path)
    for shuffle in (False, True):
        dl = DeviceNpzDataLoader(fmap, path, batch_size=96, shuffle=shuffle, device="cpu", seed=3)
        assert len(dl) == 11 and dl.num_samples == 1000
        got = {k: [] for k in full}
        for batch in dl:
            assert set(batch) == set(full)
            for k, v in batch.items():
                got[k].append(v.numpy().copy())
        label = np.concatenate(got["label"])
        assert label.shape == (1000,)
        # identify samples by a numeric column (continuous -> unique) and compare whole rows
        key = next(k for k, s in g.features.items() if s["type"] == "numeric")
        order = np.argsort(np.concatenate(got[key]), kind="stable")
        ref_order = np.argsort(full[key].astype(np.float32), kind="stable")
        for k in full:
            a = np.concatenate(got[k])[order]
            r = np.asarray(full[k])[ref_order]
            np.testing.assert_array_equal(a, r.astype(a.dtype))
        if not shuffle:
            np.testing.assert_array_equal(np.concatenate(got[key]), full[key].astype(np.float32))


@pytest.mark.gpu
def test_loader_feeds_the_model_like_host_batches(tmp_path):
    from fuxictr_amd import zoo
    from fuxictr_amd.dataloader import DeviceNpzDataLoader
    g = Golden("deepfm_adam")
    m = g.meta
    path, full = _write_npz(tmp_path, g, 640)
    fmap = _fmap(g, tmp_path)

    def make():
        torch.manual_seed(0)
        model = zoo.DeepFM(fmap, model_id="dl", gpu=0, embedding_dim=m["embedding_dim"],
                           hidden_units=m["hidden"], learning_rate=m["lr"], optimizer="adam",
                           loss="binary_crossentropy", task="binary_classification",
                           metrics=["logloss", "AUC"], verbose=0, model_root=str(tmp_path))
        model.load_state_dict({k: torch.from_numpy(v) for k, v in g.state0.items()})
        return model
    a, b = make(), make()
    dl = DeviceNpzDataLoader(fmap, path, batch_size=128, shuffle=False, device="cuda:0")
    la = []
    a.train()
    for batch in dl:
        assert all(v.is_cuda for v in batch.values())
        la.append(float(a.train_step(batch).item()))
    lb = []
    b.train()
    for i in range(0, 640, 128):
        lb.append(float(b.train_step({k: torch.from_numpy(np.asarray(v)[i:i + 128])
                                       for k, v in full.items()}).item()))
    assert la == lb                                   # same kernels, same inputs: bit-identical
    ra = a.evaluate(DeviceNpzDataLoader(fmap, path, batch_size=200, device="cuda:0"))
    rb = b.evaluate([{k: torch.from_numpy(np.asarray(v)) for k, v in full.items()}])
    assert abs(ra["logloss"] - rb["logloss"]) < 1e-9 and abs(ra["AUC"] - rb["AUC"]) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("hip_graph", [False, True])
def test_end_to_end_fit_with_device_loader(tmp_path, hip_graph):
    """The run_expid flow on the native path: device loaders -> fit (eval every 3 steps, early-stop
    bookkeeping, best checkpoint) -> evaluate -> predict -> reload; with and without hipGraph replay
    (ragged last batch included) the losses going down and the two modes agreeing."""
    from fuxictr_amd import zoo
    from fuxictr_amd.dataloader import DeviceNpzDataLoader
    g = Golden("deepfm_adam")
    m = g.meta
    rng = np.random.default_rng(11)
    from make_golden import make_batches
    w = rng.normal(size=len(g.features))

    def labelled(n):
        b = make_batches(rng, g.spec, n, 1)[0]
        z = sum(w[i] * (np.asarray(b[name], dtype=np.float64) % 7 - 3) / 3.0
                for i, name in enumerate(g.features))
        b["label"] = (rng.random(n) < 1 / (1 + np.exp(-z))).astype(np.float32)
        return b
    train, valid = labelled(1000), labelled(500)
    np.savez(str(tmp_path / "train.npz"), **train)
    np.savez(str(tmp_path / "valid.npz"), **valid)
    fmap = _fmap(g, tmp_path)
    torch.manual_seed(3)
    model = zoo.DeepFM(fmap, model_id="e2e%d" % hip_graph, gpu=0, embedding_dim=m["embedding_dim"],
                       hidden_units=m["hidden"], learning_rate=1e-2, optimizer="adam",
                       loss="binary_crossentropy", task="binary_classification",
                       metrics=["logloss", "AUC"], verbose=0, model_root=str(tmp_path),
                       eval_steps=3, hip_graph=hip_graph)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in g.state0.items()})
    tr = DeviceNpzDataLoader(fmap, str(tmp_path / "train.npz"), batch_size=128, shuffle=True,
                             device="cuda:0", seed=5)
    va = DeviceNpzDataLoader(fmap, str(tmp_path / "valid.npz"), batch_size=256, device="cuda:0")
    before = model.evaluate(va)
    model.fit(tr, epochs=3, validation_data=va)
    after = model.evaluate(va)
    assert after["logloss"] < before["logloss"] and after["AUC"] > 0.6
    pred = model.predict(va)
    assert pred.shape == (500,) and pred.dtype == np.float64 and (pred > 0).all() and (pred < 1).all()
    other = zoo.DeepFM(fmap, model_id="e2e_reload", gpu=0, embedding_dim=m["embedding_dim"],
                       hidden_units=m["hidden"], learning_rate=1e-2, optimizer="adam",
                       loss="binary_crossentropy", task="binary_classification",
                       metrics=["logloss", "AUC"], verbose=0, model_root=str(tmp_path))
    other.load_weights(model.checkpoint)
    np.testing.assert_array_equal(other.predict(va), pred)
    tmp_path.joinpath("auc_%d.txt" % hip_graph).write_text(repr(after["AUC"]))


def test_loader_keeps_meta_columns_on_the_host(tmp_path):
    """A `meta` column (group_id of the group metrics) is yielded with every batch, as the
    reference's NpzDataLoader does — BaseModel.get_group_id reads it (rank_model.py:206-208)."""
    import copy
    from fuxictr_amd.dataloader import DeviceNpzDataLoader
    from fuxictr_amd.features import FeatureMap
    g = Golden("deepfm_adam")
    path, full = _write_npz(tmp_path, g, 300)
    full = dict(full)
    full["uid"] = np.arange(300, dtype=np.int64) * 7
    np.savez(path, **full)
    spec = copy.deepcopy(g.spec)
    spec["features"].append({"uid": {"type": "meta"}})
    spec["group_id"] = "uid"
    fmap = FeatureMap(spec["dataset_id"], str(tmp_path))
    fmap.load_dict(spec, {"embedding_dim": g.meta["embedding_dim"]})
    dl = DeviceNpzDataLoader(fmap, path, batch_size=64, shuffle=True, device="cpu", seed=1)
    key = next(k for k, s in g.features.items() if s["type"] == "numeric")
    for batch in dl:
        assert "uid" in batch and not batch["uid"].is_cuda
        rows = batch["uid"].numpy() // 7                      # the samples this batch holds
        np.testing.assert_array_equal(batch[key].numpy(), full[key][rows].astype(np.float32))
