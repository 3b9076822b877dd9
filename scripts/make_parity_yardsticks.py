#!/usr/bin/env python3
"""tests/golden/parity_yardsticks.json out of scripts/parity_sweep.py's rows: per case / id distribution the RMS
over the seeds of the reference algorithm's own yardsticks (oracle with fp64 gradients, oracle on ATen's GPU
kernels) against the fp32 CPU oracle — the larger of the two.  usage: make_parity_yardsticks.py out.json rows.jsonl..."""
import json
import sys

import numpy as np

rows = []
for p in sys.argv[2:]:
    rows += [json.loads(l) for l in open(p) if l.startswith("{")]
out = {"_doc": "8-seed spread of the reference algorithm's own yardsticks (oracle with fp64 gradients, oracle on ATen's "
               "GPU kernels) against the fp32 CPU oracle at the BASELINE layer shapes: 10 Adam steps, 64 k hold-out, "
               "model seeds 1..8 (scripts/parity_sweep.py on an MI355X box, refreshed in round 5; raw rows in "
               "profiles/r05_parity_sweep_*.jsonl).  rms = root mean square over the seeds, larger of the two "
               "yardsticks; tests/baseline_shapes.py bounds |dAUC| and |dlogloss| of the native path by 3 x these "
               "(x the draw's hardness)."}
for case in sorted(set(r["case"] for r in rows)):
    for dist in sorted(set(r["dist"] for r in rows if r["case"] == case)):
        ent = {}
        for k in ("dAUC", "dLL", "mean", "loss", "max"):
            best, n = 0.0, 0
            for who in ("ref64", "refgpu"):
                sel = [r[k] for r in rows if r["case"] == case and r["dist"] == dist and r["who"] == who]
                if sel:
                    v = float(np.sqrt(np.mean(np.square(sel))))
                    if v > best:
                        best, n = v, len(sel)
            ent[k] = {"rms": best, "n": n}
        out["%s/%s" % (case, dist)] = ent
json.dump(out, open(sys.argv[1], "w"), indent=0)
print(json.dumps({k: {kk: vv["rms"] for kk, vv in v.items()} for k, v in out.items() if k != "_doc"}, indent=0))
