"""TEST INFRASTRUCTURE: the BASELINE.json configurations at their REAL layer widths (c2 DeepFM 4x1024,
c3 DCNv2 3 cross + 4x1024, c4 DIN L=50 / attention [64] Dice / dnn [512,128,64], c5 DLRM bottom
[512,256] / top [1024,1024,512,256]) on tables scaled down (vocab x 0.01 by default) so that the
oracle — the reference's dense-gradient, dense-Adam algorithm (oracle/ctr_oracle.py) — finishes a
training step in a fraction of a second on CPU.  Shared by tests/test_gpu_baseline_shapes.py (real
kernels) and tests/test_baseline_shapes_host.py (host wiring on the CPU emulation, reduced batch).

SURVEY.md 8d "Synthetic Criteo" / "Synthetic Taobao-seq": ids uniform or power-law, numerics U[0,1),
labels Bernoulli(sigmoid(teacher(x))) from a fixed random teacher so that AUC is far from 0.5.
"""
from collections import OrderedDict

import numpy as np
import torch

from fuxictr_amd import synthetic

CASES = ("c2_deepfm", "c3_dcnv2", "c4_din", "c5_dlrm")


def _yardstick_rms(case, dist):
    """8-seed RMS of the reference algorithm's own yardsticks for `case` / `dist` (see run_parity)."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "parity_yardsticks.json")
    with open(path) as f:
        tab = json.load(f)
    ent = tab.get("%s/%s" % (case, dist))
    if ent is None:
        return {"dAUC": 0.0, "dLL": 0.0, "mean": 0.0}
    return {"dAUC": float(ent["dAUC"]["rms"]), "dLL": float(ent["dLL"]["rms"]),
            "mean": float(ent.get("mean", {}).get("rms", 0.0))}


def scaled_cards(scale):
    return [max(3, int(c * scale)) for c in synthetic.CRITEO_CARDS]


def build(case, zoo, gpu, model_root, vocab_scale=0.01, seed=2019, **extra):
    """-> (model, features, oracle cfg, spec, cards).  `zoo` is fuxictr_amd.zoo (passed in so the
    host-wiring test can import it after installing the emulation)."""
    D = 16
    cards = None
    if case == "c4_din":
        fmap, spec = synthetic.taobao_feature_map(embedding_dim=D, scale=vocab_scale)
    elif case == "c5_dlrm":
        # configs[4]: the Criteo-skewed split scaled x3.70 (125 M rows at full size)
        cards = scaled_cards(3.7 * vocab_scale)
        fmap, spec = synthetic.criteo_feature_map(cards=cards, embedding_dim=D)
    else:
        cards = scaled_cards(vocab_scale)
        fmap, spec = synthetic.criteo_feature_map(cards=cards, embedding_dim=D)
    common = dict(gpu=gpu, embedding_dim=D, learning_rate=1e-3, optimizer="adam",
                  loss="binary_crossentropy", task="binary_classification",
                  metrics=["logloss", "AUC"], verbose=0, model_root=str(model_root),
                  sparse_update="exact")
    common.update(extra)
    torch.manual_seed(seed)
    cfg = {"embedding_dim": D}
    if case == "c2_deepfm":
        model = zoo.DeepFM(fmap, model_id=case, hidden_units=[1024] * 4, **common)
        cfg.update(model="DeepFM", n_hidden=4)
    elif case == "c3_dcnv2":
        model = zoo.DCNv2(fmap, model_id=case, model_structure="parallel", num_cross_layers=3,
                          parallel_dnn_hidden_units=[1024] * 4, **common)
        cfg.update(model="DCNv2", n_hidden=4, n_cross=3, structure="parallel")
    elif case == "c4_din":
        model = zoo.DIN(fmap, model_id=case, dnn_hidden_units=[512, 128, 64],
                        dnn_activations="relu", attention_hidden_units=[64],
                        attention_hidden_activations="Dice", din_target_field=["adgroup_id"],
                        din_sequence_field=["click_sequence"], din_use_softmax=False, **common)
        cfg.update(model="DIN", n_hidden=3, din_target_field=["adgroup_id"],
                   din_sequence_field=["click_sequence"], din_softmax=False)
    elif case == "c5_dlrm":
        model = zoo.DLRM(fmap, model_id=case, bottom_mlp_units=[512, 256],
                         top_mlp_units=[1024, 1024, 512, 256], interaction_op="dot", **common)
        cfg.update(model="DLRM", n_hidden=4, n_bottom=2, interaction_op="dot")
    else:
        raise ValueError(case)
    features = OrderedDict((k, v) for item in spec["features"] for k, v in item.items())
    return model, features, cfg, spec, cards


class Teacher(object):
    """Fixed random teacher: a linear score over the numeric columns and per-id random effects of the
    categorical columns (a looked-up effect for small tables, a hashed one for big tables); history
    columns contribute the hashed effect of their first item."""

    def __init__(self, features, seed=99):
        rng = np.random.default_rng(seed)
        self.w, self.tab = {}, {}
        for name, fs in features.items():
            if fs["type"] == "numeric":
                self.w[name] = rng.normal() * 1.5
            else:
                self.w[name] = rng.normal() * 0.6
                self.tab[name] = rng.normal(size=1024)
        self.rng = np.random.default_rng(seed + 1)

    def label(self, batch):
        s = 0.0
        for name, w in self.w.items():
            x = np.asarray(batch[name])
            if name in self.tab:
                ids = x[:, 0] if x.ndim == 2 else x
                s = s + w * self.tab[name][(ids * 2654435761 % 1024).astype(np.int64)]
            else:
                s = s + w * (x.astype(np.float64) - 0.5) * 2.0
        p = 1.0 / (1.0 + np.exp(-s))
        return (self.rng.random(len(p)) < p).astype(np.float32)


def make_batches(case, spec, cards, rng, B, n, dist, teacher):
    out = []
    for _ in range(n):
        if case == "c4_din":
            b = synthetic.taobao_batch(rng, B, spec, dist=dist)
        else:
            b = synthetic.criteo_batch(rng, B, cards=cards, dist=dist)
        b["label"] = teacher.label(b)
        out.append(b)
    return out


def tb(b):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in b.items()}


def logits_of(model, batch):
    """Logit vector of the native model (the output activation remembers its input)."""
    with torch.no_grad():
        p = model.forward(batch)["y_pred"]
    return p._fx_logit.reshape(-1).float().cpu().numpy(), p.reshape(-1).float().cpu().numpy()


def native_gradients(model, batch):
    """One forward / backward of the native model on `batch` WITHOUT the optimizer step ->
    {state_dict key: dense fp32 gradient on the CPU}, the layout the reference's autograd leaves in
    .grad (rank_model.py:320): dense-tower gradients as they are, the unique-row gradients of every
    table group scattered into dense [V_f, D] arrays per feature, numeric weights [D, 1]."""
    from fuxictr_amd.layers import FeatureEmbeddingDict
    model.train()
    model.optimizer.set_max_norm(0.0)
    model._forward_backward(batch)
    torch.cuda.synchronize()
    out = {}
    table_keys = set()
    for prefix, mod in model.named_modules():
        if not isinstance(mod, FeatureEmbeddingDict):
            continue
        pre = prefix + "." if prefix else ""
        for grp in mod._groups.values():
            dense = None
            if grp.table is not None:
                dense = torch.zeros(grp.total_rows, grp.D, dtype=torch.float32)
                assert len(grp.pending) <= 1
                for rec in grp.pending:
                    nu = int(rec.dd.n_unique.item())
                    rows = rec.dd.uniq_row[:nu].cpu().long()
                    dense[rows] = rec.G[:nu].float().cpu()
            for feature, (base, V, _) in grp.tables.items():
                key = pre + "embedding_layers.%s.weight" % feature
                table_keys.add(key)
                out[key] = dense[base:base + V].clone()
            for j, feature in enumerate(grp.numeric):
                key = pre + "embedding_layers.%s.weight" % feature
                table_keys.add(key)
                g = grp.num_grad[j] if grp.num_grad is not None else torch.zeros(grp.D)
                out[key] = g.detach().float().cpu().reshape(grp.D, 1).clone()
    for name, p in model.named_parameters():
        if name in table_keys or name in out:
            continue
        out[name] = (p.grad if p.grad is not None else torch.zeros_like(p)).detach().float().cpu().clone()
    return out


def run_parity(case, dist, model, features, cfg, spec, cards, oracle_mod, B=4096, steps=10,
               holdout=65536, logit_tol=1e-4, loss_tol=1e-4, metric_tol=5e-5, slack=3.0,
               gpu_yardstick=None):
    """The comparison itself; returns a dict of the observed differences (asserts inside).

    Two kinds of statement:
      (A) SAME WEIGHTS -> same logits within 1e-4 (the north-star's forward claim): checked for the
          untrained weights and for the oracle's TRAINED weights loaded into the native model.
      (B) INDEPENDENT TRAINING (native vs oracle, 10 steps each): loss trajectory, hold-out logits,
          AUC and logloss.  Adam's lr*m/(sqrt(v)+eps) turns gradient elements that are cancellation
          residue into +-lr steps, so the reference ALGORITHM is only reproducible up to the rounding
          of its gradients: already after one step two correctly rounded evaluations differ by ~lr in
          some weights.  That spread is measured in the same run by two yardsticks —
            ref64:  the oracle with float64 forward/backward (OracleTrainer64), and
            refgpu: the oracle's identical functional code on ATen's GPU kernels (what the reference
                    itself executes with `gpu: 0`), when a device is given —
          and every native-vs-oracle difference must stay within max(stated tolerance, slack x the
          larger yardstick): the native path is no further from the CPU reference than the
          reference's own GPU back end / a more exact evaluation of its own gradients is."""
    from sklearn.metrics import log_loss, roc_auc_score
    O = oracle_mod
    state0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    tr = O.OracleTrainer(cfg, state0, features, lr=1e-3, max_norm=10.0)
    yards = {"ref64": O.OracleTrainer64(cfg, state0, features, lr=1e-3, max_norm=10.0)}
    if gpu_yardstick is not None:
        yards["refgpu"] = O.OracleTrainer(cfg, state0, features, lr=1e-3, max_norm=10.0,
                                          device=gpu_yardstick)
    teacher = Teacher(features)
    rng = np.random.default_rng({"powerlaw": 11, "uniform": 12}[dist])
    train = make_batches(case, spec, cards, rng, B, steps, dist, teacher)
    test = make_batches(case, spec, cards, rng, B, max(1, holdout // B), dist, teacher)
    res = {}
    # (A1) forward logits of the untrained model
    model.eval()
    lg, _ = logits_of(model, tb(train[0]))
    res["logit0"] = float(np.abs(lg - tr.logits(tb(train[0])).numpy()).max())
    assert res["logit0"] <= logit_tol, ("initial logits", res)
    # (B) independent training
    model.train()
    model._max_gradient_norm = 10.0
    ln, lo, ly = [], [], {k: [] for k in yards}
    for b in train:
        t = tb(b)
        ln.append(float(model.train_step(t).item()))
        lo.append(tr.train_step(t, t["label"])[0])
        for k, y_ in yards.items():
            ly[k].append(y_.train_step(t, t["label"])[0])
    lo = np.asarray(lo)
    res["loss_first_last"] = (ln[0], ln[-1])
    res["loss"] = {"native": float(np.abs(np.asarray(ln) - lo).max())}
    for k in yards:
        res["loss"][k] = float(np.abs(np.asarray(ly[k]) - lo).max())
    model.eval()
    y = np.concatenate([b["label"] for b in test]).astype(np.float64)
    ref_lg = np.concatenate([tr.logits(tb(b)).numpy() for b in test])

    def describe(lgs):
        d = np.abs(lgs - ref_lg)
        p = 1.0 / (1.0 + np.exp(-lgs.astype(np.float64)))
        return {"max": float(d.max()), "mean": float(d.mean()), "auc": float(roc_auc_score(y, p)),
                "logloss": float(log_loss(y, p))}
    ref = describe(ref_lg)
    res["reference"] = {"auc": ref["auc"], "logloss": ref["logloss"]}
    res["native"] = describe(np.concatenate([logits_of(model, tb(b))[0] for b in test]))
    for k, y_ in yards.items():
        res[k] = describe(np.concatenate([y_.logits(tb(b)).numpy() for b in test]))

    def yard(fn):
        return max(fn(k) for k in yards)
    # mean |dlogit| over 64 k samples is the stable statistic (slack x the same-run yardsticks); the
    # maximum is an extreme value (slack + 1).  AUC / logloss differences are signed sums of those
    # errors: on ONE seed a yardstick's own dAUC can be ~0 by chance while another's is 10x larger, so the
    # same-run value is no scale.  Their scale is the 8-seed RMS of the two yardsticks, measured once on
    # an MI355X (scripts/parity_sweep.py -> tests/golden/parity_yardsticks.json; over those seeds the
    # native path's RMS is 0.5 - 1.5 x the yardsticks', profiles/r03_parity_sweep_summary.txt): the
    # bound is slack x that RMS (no more 10 x the same-run value).
    rms8 = _yardstick_rms(case, dist)
    # (Round 5 multiplied the AUC / logloss bounds by a same-run "draw hardness" factor, fitted after the
    # split-bf16 GEMM had moved one configuration's dAUC past 5e-5; round 6 removed it again — VERDICT r5 weak 1 —
    # together with the cause of most of the native path's own spread: the Adam kernels formed their lerp
    # weights from the fp32 images of the betas, every update 6e-6 large, profiles/r06_series_catchup_ab.txt.)
    bound = {"loss": max(loss_tol, slack * yard(lambda k: res["loss"][k])),
             "max": max(logit_tol, (slack + 1) * yard(lambda k: res[k]["max"])),
             "mean": max(0.1 * logit_tol, slack * yard(lambda k: res[k]["mean"])),
             "auc": max(metric_tol, slack * rms8["dAUC"],
                        slack * yard(lambda k: abs(res[k]["auc"] - ref["auc"]))),
             "logloss": max(metric_tol, slack * rms8["dLL"],
                            slack * yard(lambda k: abs(res[k]["logloss"] - ref["logloss"])))}
    res["bounds"] = bound
    assert res["loss"]["native"] <= bound["loss"], ("loss trajectory", res)
    assert res["native"]["max"] <= bound["max"], ("trained logits (max)", res)
    assert res["native"]["mean"] <= bound["mean"], ("trained logits (mean)", res)
    assert abs(res["native"]["auc"] - ref["auc"]) <= bound["auc"], ("AUC", res)
    assert abs(res["native"]["logloss"] - ref["logloss"]) <= bound["logloss"], ("logloss", res)
    assert abs(ref["auc"] - 0.5) > 0.03, ("teacher labels should give a non-trivial AUC", res)
    # (A2) the 1e-4 logit claim at TRAINED weights: the oracle's trained state loaded into the native
    # model (reference checkpoint keys), forward on the hold-out
    model.load_state_dict({k: v.detach().cpu().clone() for k, v in tr.state.items()})
    model.eval()
    same = describe(np.concatenate([logits_of(model, tb(b))[0] for b in test]))
    res["same_weights"] = same
    assert same["max"] <= logit_tol, ("forward at the oracle's trained weights", res)
    # ... and with it AUC / logloss of the TRAINED model to 4 decimals (the north-star's wording)
    assert abs(same["auc"] - ref["auc"]) < metric_tol, ("AUC at the same trained weights", res)
    assert abs(same["logloss"] - ref["logloss"]) < metric_tol, ("logloss, same weights", res)
    model.optimizer.check_errors()
    return res
