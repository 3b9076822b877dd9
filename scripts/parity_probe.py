#!/usr/bin/env python3
"""Spread of the baseline-shape parity quantities over model seeds: native vs oracle next to the two
yardsticks (oracle with fp64 gradients, oracle on ATen GPU kernels).  usage: parity_probe.py case dist seeds..."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import baseline_shapes as BS  # noqa: E402
from fuxictr_amd import zoo  # noqa: E402
from oracle import ctr_oracle as O  # noqa: E402

case, dist = sys.argv[1], sys.argv[2]
for seed in [int(x) for x in sys.argv[3:]]:
    model, features, cfg, spec, cards = BS.build(case, zoo, 0, "/tmp/fx_probe", seed=seed)
    try:
        res = BS.run_parity(case, dist, model, features, cfg, spec, cards, O, gpu_yardstick="cuda:0",
                            slack=1e9)
    except AssertionError as e:
        res = e.args[0][1] if isinstance(e.args[0], tuple) else {"error": str(e)[:300]}
    ref = res.get("reference", {})
    row = {"case": case, "dist": dist, "seed": seed, "loss": res.get("loss")}
    for k in ("native", "ref64", "refgpu"):
        if k in res:
            row[k] = {"max": res[k]["max"], "mean": res[k]["mean"],
                      "dAUC": res[k]["auc"] - ref["auc"], "dLL": res[k]["logloss"] - ref["logloss"]}
    print(json.dumps(row), flush=True)
    del model
