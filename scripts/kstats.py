#!/usr/bin/env python3
"""Compact summary of a rocprofv3 *_kernel_stats.csv: short kernel names, per-step microseconds."""
import csv
import re
import sys

path, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(path)))
tot = 0.0
out = []
for r in rows:
    name = r["Name"]
    m = re.search(r"(k_[a-z0-9_]+|radix_sort[a-z_]*|merge_sort[a-z_]*|scan_impl|init_lookback[a-z_]*|"
                  r"transform_impl|CatArray|FillFunctor|copyBuffer|normal_kernel|reduce_kernel|"
                  r"vectorized_elementwise_kernel|Cijk[A-Za-z0-9_]*)", name)
    short = m.group(1) if m else name[:40]
    tmpl = re.search(r"(k_[a-z0-9_]+<[^>]*>)", name)
    if tmpl:
        short = tmpl.group(1)
    t = float(r["TotalDurationNs"]) / 1e3
    out.append((t, short, int(r["Calls"])))
    if "copyBuffer" not in short and "normal_kernel" not in short:
        tot += t
for t, s, c in sorted(out, reverse=True)[: int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
    print("%10.1f us/step  %6.1f calls/step  %s" % (t / steps, c / steps, s))
print("total (no copies/init): %.1f us/step" % (tot / steps))
