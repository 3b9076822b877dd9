"""BASELINE.json configs[0] (plumbing): the reference's demo `DeepFM_test_npz`
(demo/config/example3_config, data/tiny_npz, embedding_regularizer 1e-8, one epoch = one step)
replayed through the native BaseModel.fit / evaluate.  tests/golden/c1_tiny_npz.npz holds the
batches the reference's own RankDataLoader fed to fit(), its weights before/after and its
validation logloss / AUC (0.6798385031 / 0.9661458333 — the known answers of SURVEY.md §8c).
The same for the model zoo's own smoke configurations (`DIN_test` on data/tiny_seq: 0.6878952344 /
0.796875; `xDeepFM_test`: 0.6672440633 / 0.9869791667; `DLRM_test`: 0.6888397238 / 0.65234375;
`DCNv2_test`: 0.6797173729 / 1.0 — the last three with data/tiny_npz in place of data/tiny_parquet,
see tests/golden/make_golden.py:DEMOS)."""
import numpy as np
import pytest
import torch

import _cpu_emul
from conftest import Golden, assert_weights_close
from oracle import ctr_oracle as O
from test_host_wiring import _cpu_opt_init, tb


class Gen(list):
    pass


DEMOS = ["c1_tiny_npz", "demo_din_tiny_seq", "demo_xdeepfm_tiny_npz", "demo_dlrm_tiny_npz",
         "demo_dcnv2_tiny_npz"]      # (slim fixture: the trained 560x560 cross weights as a row sample)


def _golden(name):
    import os
    from conftest import ROOT
    if not os.path.exists(os.path.join(ROOT, "tests", "golden", name + ".npz")):
        pytest.skip("fixture %s.npz not generated (tests/golden/make_golden.py %s)" % (name, name))
    return Golden(name)


def _model(g, tmp_path, gpu, **kw):
    from fuxictr_amd import zoo
    from fuxictr_amd.features import FeatureMap
    m = g.meta
    fmap = FeatureMap(g.spec["dataset_id"], str(tmp_path))
    fmap.load_dict(g.spec, {"embedding_dim": m["embedding_dim"]})
    common = dict(model_id=m.get("expid") or m["name"], gpu=gpu, embedding_dim=m["embedding_dim"],
                  learning_rate=m["lr"], optimizer=m["optimizer"], loss="binary_crossentropy",
                  task="binary_classification", metrics=["logloss", "AUC"], verbose=0,
                  model_root=str(tmp_path), embedding_regularizer=m["emb_reg"],
                  net_regularizer=m["net_reg"], **kw)
    if m["model"] == "DeepFM":
        model = zoo.DeepFM(fmap, hidden_units=m["hidden"], **common)
    elif m["model"] == "DIN":
        model = zoo.DIN(fmap, dnn_hidden_units=m["hidden"], dnn_activations="relu",
                        attention_hidden_units=m["att_hidden"],
                        attention_hidden_activations="Dice", din_target_field="adgroup_id",
                        din_sequence_field="click_sequence", **common)
    elif m["model"] == "xDeepFM":
        model = zoo.xDeepFM(fmap, dnn_hidden_units=m["hidden"], cin_hidden_units=m["cin"], **common)
    elif m["model"] == "DLRM":
        model = zoo.DLRM(fmap, top_mlp_units=m["hidden"], bottom_mlp_units=m["bottom"],
                         interaction_op="dot", **common)
    else:
        model = zoo.DCNv2(fmap, model_structure=m.get("structure", "parallel"), num_cross_layers=m["n_cross"],
                          parallel_dnn_hidden_units=m["hidden"],
                          stacked_dnn_hidden_units=m.get("stacked", []), **common)
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
    pred = model.predict(valid)
    np.testing.assert_allclose(pred, g.expect["pred1"], atol=1e-6)
    # AUC to 4 decimals — except for (positive, negative) pairs the reference itself separates by
    # less than the prediction tolerance: after ONE step from a 1e-4 init all 100 predictions sit
    # within ~1e-4 of 0.5, and such a pair may order either way under another fp32 summation order
    # (each flip moves AUC by 1 / (n_pos * n_neg) = 0.0013 here)
    y = np.asarray(g.batches[-1][g.spec["labels"][0]]).reshape(-1)
    ref = np.asarray(g.expect["pred1"], dtype=np.float64)
    pos, neg = ref[y > 0.5], ref[y <= 0.5]
    near_ties = int((np.abs(pos[:, None] - neg[None, :]) < 2e-6).sum())
    slack = near_ties / float(len(pos) * len(neg))
    if near_ties == 0:
        assert round(logs["AUC"], 4) == round(float(g.expect["valid_auc"][0]), 4)
    else:
        assert abs(logs["AUC"] - float(g.expect["valid_auc"][0])) <= slack + 1e-9, (logs, slack)
    sd = {k: v.cpu().numpy() for k, v in model.state_dict().items()}
    for k, got, ref in g.final_weights(sd):
        assert_weights_close(got, ref, m["lr"], m["steps"], k, tol=1e-6)


@pytest.mark.parametrize("demo", DEMOS)
def test_oracle_reproduces_the_reference_demo_run(demo):
    g = _golden(demo)
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
    for k, got, ref in g.final_weights({k: v.detach().numpy() for k, v in tr.state.items()}):
        assert_weights_close(got, ref, m["lr"], m["steps"], k, tol=1e-6)


@pytest.mark.parametrize("demo", DEMOS)
def test_c1_host_wiring_fit_evaluate(demo, tmp_path, monkeypatch):
    _cpu_emul.install(monkeypatch)
    from fuxictr_amd import optim
    monkeypatch.setattr(optim._NativeOptimizer, "__init__", _cpu_opt_init(optim))
    g = _golden(demo)
    _fit_and_check(_model(g, tmp_path, gpu=-1), g)


@pytest.mark.gpu
@pytest.mark.parametrize("hip_graph", [False, True])
@pytest.mark.parametrize("demo", DEMOS)
def test_c1_native_fit_evaluate_matches_the_reference_demo(demo, tmp_path, hip_graph):
    g = _golden(demo)
    _fit_and_check(_model(g, tmp_path, gpu=0, hip_graph=hip_graph), g)


class EpochGen(object):
    """A training 'loader' that replays the batches the reference's shuffling DataLoader produced:
    the e-th iteration yields epoch e's batches."""

    def __init__(self, epochs):
        self.epochs, self.it = epochs, 0

    def __len__(self):
        return len(self.epochs[0])

    def __iter__(self):
        e = self.epochs[min(self.it, len(self.epochs) - 1)]
        self.it += 1
        return iter(e)


def _fit_control(model, g):
    """BaseModel.fit over several epochs == the reference's run of the same configuration:
    every in-fit evaluation, the learning rate after each one (reduce-on-plateau: x0.1, three times),
    the early stop after `early_stop_patience` stalls, and the reload of the best checkpoint."""
    m = g.meta
    spe = m["steps_per_epoch"]
    steps = [tb(b) for b in g.batches[:-1]]
    epochs = [steps[i:i + spe] for i in range(0, len(steps), spe)]
    valid = Gen([tb(g.batches[-1])])
    evals, lrs = [], []
    inner_eval, inner_ckpt = model.evaluate, model.checkpoint_and_earlystop

    def rec_eval(gen, metrics=None, **kw):
        logs = inner_eval(gen, metrics=metrics, **kw)
        evals.append([float(logs["logloss"]), float(logs["AUC"])])
        return logs

    def rec_ckpt(logs, **kw):
        inner_ckpt(logs, **kw)
        lrs.append(float(model.optimizer.param_groups[0]["lr"]))
    model.evaluate, model.checkpoint_and_earlystop = rec_eval, rec_ckpt
    model.fit(EpochGen(epochs), epochs=m["epochs"], validation_data=valid)
    model.evaluate, model.checkpoint_and_earlystop = inner_eval, inner_ckpt
    assert len(evals) == len(g.expect["fit_evals"]) == len(epochs)        # stopped at the same epoch
    np.testing.assert_allclose(lrs, g.expect["fit_lrs"], rtol=1e-6)
    np.testing.assert_allclose(np.asarray(evals)[:, 0], g.expect["fit_evals"][:, 0], atol=2e-5)
    np.testing.assert_allclose(np.asarray(evals)[:, 1], g.expect["fit_evals"][:, 1], atol=1e-4)
    logs = model.evaluate(valid)                                            # the reloaded best weights
    assert abs(logs["logloss"] - float(g.expect["valid_logloss"][0])) <= 2e-5
    pred = model.predict(valid)
    np.testing.assert_allclose(pred, g.expect["pred1"], atol=2e-5)
    sd = model.state_dict()
    for k, ref in g.state1.items():
        assert_weights_close(sd[k].cpu().numpy(), ref, 0.02, len(steps), k, tol=5e-5)


def _fit_control_model(g, tmp_path, gpu, **kw):
    m = g.meta
    return _model(g, tmp_path, gpu, early_stop_patience=m["early_stop_patience"],
                  monitor=m["monitor"], **kw)


def test_fit_control_loop_matches_the_reference(tmp_path, monkeypatch):
    _cpu_emul.install(monkeypatch)
    from fuxictr_amd import optim
    monkeypatch.setattr(optim._NativeOptimizer, "__init__", _cpu_opt_init(optim))
    g = Golden("fit_control_tiny_npz")
    _fit_control(_fit_control_model(g, tmp_path, gpu=-1), g)


@pytest.mark.gpu
def test_fit_control_loop_matches_the_reference_on_gpu(tmp_path):
    g = Golden("fit_control_tiny_npz")
    _fit_control(_fit_control_model(g, tmp_path, gpu=0), g)
