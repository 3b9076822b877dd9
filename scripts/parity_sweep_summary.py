#!/usr/bin/env python3
"""Table of a parity sweep (scripts/parity_sweep.py *.jsonl): per case / id distribution / evaluation, over the
model seeds: mean |dlogit| (rms, median, max), |dAUC| (rms, median), loss-trajectory difference (rms) — every
figure against the fp32 CPU oracle of the same seed.  usage: parity_sweep_summary.py a.jsonl b.jsonl ..."""
import json
import sys

import numpy as np

rows = []
for path in sys.argv[1:]:
    with open(path) as f:
        rows += [json.loads(l) for l in f if l.strip()]
print("%-10s %-9s %-18s %2s | mean |dlogit| rms/median/max | |dAUC| rms/median       | loss-trajectory rms"
      % ("case", "dist", "who", "n"))
for case in sorted(set(r["case"] for r in rows)):
    for dist in ("powerlaw", "uniform"):
        sel = [r for r in rows if r["case"] == case and r["dist"] == dist]
        for who in sorted(set(r["who"] for r in sel)):
            s = [r for r in sel if r["who"] == who]
            m = np.array([r["mean"] for r in s])
            a = np.abs(np.array([r["dAUC"] for r in s]))
            lo = np.array([r["loss"] for r in s])
            print("%-10s %-9s %-18s %2d | %.2e %.2e %.2e | %.2e %.2e     | %.2e"
                  % (case, dist, who, len(s), np.sqrt((m ** 2).mean()), np.median(m), m.max(),
                     np.sqrt((a ** 2).mean()), np.median(a), np.sqrt((lo ** 2).mean())))
        print()
