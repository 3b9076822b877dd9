"""BASELINE.json configs[0] (plumbing): the reference's demo `DeepFM_test_npz`
(demo/config/example3_config, data/tiny_npz, embedding_regularizer 1e-8, one epoch = one step)
replayed through the native BaseModel.fit / evaluate.  tests/golden/c1_tiny_npz.npz holds the
batches the reference's own RankDataLoader fed to fit(), its weights before/after and its
validation logloss / AUC (0.6798385031 / 0.9661458333 — the known answers of SURVEY.md §8c)."""
import numpy as np
import pytest
import torch

import _cpu_emul
from conftest import Golden, assert_weights_close
from oracle import ctr_oracle as O
from test_host_wiring import _cpu_opt_init, tb


class Gen(list):
    pass


def _model(g, tmp_path, gpu, **kw):
    from fuxictr_amd import zoo
    from fuxictr_amd.features import FeatureMap
    m = g.meta
    fmap = FeatureMap(g.spec["dataset_id"], str(tmp_path))
    fmap.load_dict(g.spec, {"embedding_dim": m["embedding_dim"]})
    model = zoo.DeepFM(fmap, model_id="DeepFM_test_npz", gpu=gpu, embedding_dim=m["embedding_dim"],
                       hidden_units=m["hidden"], learning_rate=m["lr"], optimizer=m["optimizer"],
                       loss="binary_crossentropy", task="binary_classification",
                       metrics=["logloss", "AUC"], verbose=0, model_root=str(tmp_path),
                       embedding_regularizer=m["emb_reg"], net_regularizer=m["net_reg"], **kw)
    sd = {k: torch.from_numpy(v) for k, v in g.state0.items()}
    assert sorted(model.state_dict().keys()) == sorted(sd.keys())
    model.load_state_dict(sd)
    return model


def _fit_and_check(model, g):
    m = g.meta
    train = Gen(tb(b) for b in g.batches[:-1])
    valid = Gen([tb(g.batches[-1])])
    model.fit(train, epochs=1, validation_data=valid)
    logs = model.evaluate(valid)
    assert abs(logs["logloss"] - float(g.expect["valid_logloss"][0])) <= 1e-6
    assert round(logs["AUC"], 4) == round(float(g.expect["valid_auc"][0]), 4)
    pred = model.predict(valid)
    np.testing.assert_allclose(pred, g.expect["pred1"], atol=1e-6)
    sd = model.state_dict()
    for k, ref in g.state1.items():
        assert_weights_close(sd[k].cpu().numpy(), ref, m["lr"], m["steps"], k, tol=1e-6)


def test_oracle_reproduces_the_reference_demo_run():
    g = Golden("c1_tiny_npz")
    m = g.meta
    label = g.spec["labels"][0]
    tr = O.OracleTrainer(g.cfg(), g.state0, g.features, lr=m["lr"], max_norm=m["max_norm"],
                         optimizer=m["optimizer"], emb_reg=O.parse_regularizer(m["emb_reg"]),
                         net_reg=O.parse_regularizer(m["net_reg"]))
    losses = []
    for b in g.batches[:-1]:
        b = tb(b)
        losses.append(tr.train_step(b, b[label])[0])
    np.testing.assert_allclose(losses, g.expect["loss"], atol=1e-6)
    p = tr.predict(tb(g.batches[-1])).reshape(-1).numpy()
    np.testing.assert_allclose(p, g.expect["pred1"], atol=1e-6)
    from sklearn.metrics import log_loss, roc_auc_score
    y = g.batches[-1][label]
    assert abs(log_loss(y, p.astype(np.float64)) - float(g.expect["valid_logloss"][0])) <= 1e-6
    assert round(roc_auc_score(y, p), 4) == round(float(g.expect["valid_auc"][0]), 4)
    for k, ref in g.state1.items():
        assert_weights_close(tr.state[k].detach().numpy(), ref, m["lr"], m["steps"], k, tol=1e-6)


def test_c1_host_wiring_fit_evaluate(tmp_path, monkeypatch):
    _cpu_emul.install(monkeypatch)
    from fuxictr_amd import optim
    monkeypatch.setattr(optim._NativeOptimizer, "__init__", _cpu_opt_init(optim))
    g = Golden("c1_tiny_npz")
    _fit_and_check(_model(g, tmp_path, gpu=-1), g)


@pytest.mark.gpu
@pytest.mark.parametrize("hip_graph", [False, True])
def test_c1_native_fit_evaluate_matches_the_reference_demo(tmp_path, hip_graph):
    g = Golden("c1_tiny_npz")
    _fit_and_check(_model(g, tmp_path, gpu=0, hip_graph=hip_graph), g)
