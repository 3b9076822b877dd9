#!/usr/bin/env python3
"""Per-tensor table out of scripts/grad_parity.py's JSON lines: relative L2 error of the native /
cpu32 / gpu32 gradients against the float64 gradient, tables aggregated (root of the summed squares)."""
import json
import sys
from collections import OrderedDict

rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
for r in rows:
    print("== %s %s seed %d [%s]" % (r["case"], r["dist"], r["seed"], r["tag"]))
    print("%-44s %10s %11s %11s %11s %7s   %s" % ("tensor", "|g|", "native", "cpu32", "gpu32", "ratio",
                                                   "max err in ulp(max|g|): native / cpu32 / gpu32"))
    agg = OrderedDict()
    for k, t in r["tensors"].items():
        name = k
        if ".embedding_layers." in k:
            name = k.split(".embedding_layers.")[0] + ".embedding_layers.* (%s)" % (
                "numeric" if t["numel"] <= 64 else "tables")
        a = agg.setdefault(name, {"n": 0.0, "native": 0.0, "cpu32": 0.0, "gpu32": 0.0, "ulp": [0, 0, 0]})
        a["n"] += t["norm"] ** 2
        for i, v in enumerate(("native", "cpu32", "gpu32")):
            a[v] += (t[v]["rel_l2"] * t["norm"]) ** 2
            a["ulp"][i] = max(a["ulp"][i], t[v]["max_ulp"])
    for name, a in agg.items():
        n = a["n"] ** 0.5
        e = [(a[v] ** 0.5) / n if n > 0 else 0.0 for v in ("native", "cpu32", "gpu32")]
        yard = max(e[1], e[2], 2e-7)
        print("%-44s %10.3e %11.3e %11.3e %11.3e %7.2f   %.1f / %.1f / %.1f" %
              (name[-44:], n, e[0], e[1], e[2], e[0] / yard, a["ulp"][0], a["ulp"][1], a["ulp"][2]))
