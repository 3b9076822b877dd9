#!/usr/bin/env python3
"""Per-tensor table out of scripts/grad_parity.py's JSON lines: relative L2 error of the native /
cpu32 / gpu32 gradients against the float64 gradient — plain, and against the float64 gradient on the
evaluation's own ReLU decisions — tables aggregated (root of the summed squares)."""
import json
import sys
from collections import OrderedDict

rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
for r in rows:
    print("== %s %s seed %d [%s]" % (r["case"], r["dist"], r["seed"], r["tag"]))
    print("   hidden units whose ReLU decision differs from the fp64 evaluation, per layer: %s"
          % r.get("relu_flips_vs_fp64_per_layer"))
    print("%-42s %9s | %9s %9s %9s %6s | %9s %9s %9s %6s" %
          ("tensor (rel. L2 error vs fp64)", "|g|", "native", "cpu32", "gpu32", "ratio", "nat/same", "cpu/same",
           "gpu/same", "ratio"))
    agg = OrderedDict()
    for k, t in r["tensors"].items():
        name = k
        if ".embedding_layers." in k:
            name = k.split(".embedding_layers.")[0] + ".embedding_layers.* (%s)" % (
                "numeric" if t["numel"] <= 64 else "tables")
        a = agg.setdefault(name, {"n": 0.0, "p": [0.0] * 3, "s": [0.0] * 3})
        a["n"] += t["norm"] ** 2
        for i, v in enumerate(("native", "cpu32", "gpu32")):
            a["p"][i] += (t[v]["rel_l2"] * t["norm"]) ** 2
            sm = t.get("same_masks", {}).get(v)
            if sm is not None:
                a["s"][i] += (sm["rel_l2"] * t["norm"]) ** 2
    for name, a in agg.items():
        n = a["n"] ** 0.5
        p = [(x ** 0.5) / n if n > 0 else 0.0 for x in a["p"]]
        s = [(x ** 0.5) / n if n > 0 else 0.0 for x in a["s"]]
        print("%-42s %9.2e | %9.2e %9.2e %9.2e %6.2f | %9.2e %9.2e %9.2e %6.2f" %
              (name[-42:], n, p[0], p[1], p[2], p[0] / max(p[1], p[2], 1e-6), s[0], s[1], s[2],
               s[0] / max(s[1], s[2], 1e-6)))
