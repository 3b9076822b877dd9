"""A 400-step native-vs-oracle trajectory at the BASELINE shapes (SURVEY.md section 4 (3):
"N-step trajectory + AUC on >= 100 k eval rows").  c2 DeepFM and c3 DCNv2 with the tables x 0.01, power-law ids
(cold rows return after hundreds of steps: the exact-mode catch-up — the Adam series table — is what brings
them up to date), 131 072 hold-out rows, the native model under hipGraph replay.  bench.parity_long_horizon
does the work (the bench line carries the same object as parity_full_vocab.steps_400).

  same weights (the oracle's state loaded into a native model at steps 100 / 200 / 400):
        logits within 1e-4, AUC and logloss within 5e-5           — unconditional
  independent training, every one of the 400 steps:
        |loss_native - loss_oracle| <= max(1e-4, 3 x what the reference's own GPU back end differs from its CPU
        back end on the same steps), hold-out mean |dlogit| / dAUC / dlogloss likewise — the SAME-RUN yardstick
        only, no fitted factor
"""
import json
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


@pytest.mark.parametrize("case", ["c2_deepfm", "c3_dcnv2"])
def test_400_steps_against_the_oracle(case):
    import bench
    r = bench.parity_long_horizon(case, 0, steps=400)
    print("[long horizon] %s: %s" % (case, json.dumps(r)))
    assert r["steps"] == 400 and r["holdout_rows"] >= 131072
    # cold rows really were far behind the optimizer when they were read again
    assert max(g["behind_max"] for g in r["row_age"]) >= 300, r["row_age"]
    assert r["loss_first_last"][1] < r["loss_first_last"][0] - 0.01, "the model should learn"
    for t in ("100", "200", "400"):
        c = r["checkpoints"][t]
        assert abs(c["reference"]["auc"] - 0.5) > 0.03, c
        s = c["same_weights"]
        assert s["max_dlogit"] <= 1e-4 and s["dauc"] < 5e-5 and s["dlogloss"] < 5e-5, (t, s)
    # the independently trained model, evaluated once at the end (an evaluation flushes the exact mode)
    c = r["checkpoints"]["400"]
    n, yd = c["independent_native"], c["independent_yardstick"]
    assert n["mean_dlogit"] <= max(1e-5, 3.0 * yd["mean_dlogit"]), (n, yd)
    assert n["dauc"] <= max(5e-5, 3.0 * yd["dauc"]), (n, yd)
    assert n["dlogloss"] <= max(5e-5, 3.0 * yd["dlogloss"]), (n, yd)
    for i, (dn, dy) in enumerate(r["dloss_by_100"]):
        assert dn <= max(1e-4, 3.0 * dy), ("steps %d..%d" % (100 * i + 1, 100 * i + 100), dn, dy)


def test_c4_din_400_steps_against_the_oracle():
    """c4 (DIN, 50-position click sequence aliasing adgroup_id, Dice attention) over the same horizon: the id
    plan that takes the bucketed in-LDS de-dup (round 6) and, with it, the row record and the series catch-up —
    400 steps at B = 1024 (the oracle's attention on the CPU is what bounds the batch), rows more than 200 steps
    behind (the x 0.01 tables are small: every row returns within ~ 240 steps), same bounds as above."""
    import bench
    r = bench.parity_long_horizon("c4_din", 0, steps=400, B=1024, checkpoints=(100, 200, 400))
    print("[long horizon] c4_din: %s" % json.dumps(r))
    assert r["steps"] == 400 and r["holdout_rows"] >= 131072
    assert max(g["behind_max"] for g in r["row_age"]) >= 200, r["row_age"]
    for t in ("100", "200", "400"):
        s = r["checkpoints"][t]["same_weights"]
        assert s["max_dlogit"] <= 1e-4 and s["dauc"] < 5e-5 and s["dlogloss"] < 5e-5, (t, s)
    c = r["checkpoints"]["400"]
    n, yd = c["independent_native"], c["independent_yardstick"]
    assert n["mean_dlogit"] <= max(1e-5, 3.0 * yd["mean_dlogit"]), (n, yd)
    # Two independently trained c4 models are ~ 0.02 apart in the logits after 400 steps — the reference's own two
    # back ends as much as the native path (mean_dlogit above).  AUC and logloss of such a pair are small SIGNED
    # differences of means: one yardstick sample of them can come out anywhere between 5e-6 and 2e-4 (both were
    # seen), so a multiple of that one sample is not a bound.  What bounds them is the perturbation itself:
    # |d logloss / d logit| <= 1, so a mean logit distance of delta moves logloss by at most delta; the test
    # allows 1 % of the YARDSTICK's delta (never the native path's own).
    slack = 0.01 * yd["mean_dlogit"]
    assert n["dauc"] <= max(5e-5, 3.0 * yd["dauc"], slack), (n, yd)
    assert n["dlogloss"] <= max(5e-5, 3.0 * yd["dlogloss"], slack), (n, yd)
    for i, (dn, dy) in enumerate(r["dloss_by_100"]):
        assert dn <= max(1e-4, 3.0 * dy), (i, dn, dy)
