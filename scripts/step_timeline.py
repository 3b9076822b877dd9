#!/usr/bin/env python3
"""One steady-state training step out of a rocprofv3 *_kernel_trace.csv: kernel, duration, gap to the
previous kernel's end.  usage: step_timeline.py trace.csv [which_step_from_the_end=2]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step starts with the batch's pack launch (k_pack_columns_multi, outside the graph); older traces:
# with the optimizer's begin-step kernel
marker = "k_pack_columns_multi" if any("k_pack_columns_multi" in r["Kernel_Name"] for r in rows) \
    else "k_opt_begin_step"
starts = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
lo, hi = starts[-back - 1], starts[-back]
step = rows[lo:hi]
t0 = int(step[0]["Start_Timestamp"])
prev_end = None
tot_k = tot_gap = 0.0
agg = {}
for r in step:
    name = r["Kernel_Name"]
    m = re.search(r"(k_[a-z0-9_]+(<[^>]*>)?|[A-Za-z_]*elementwise[A-Za-z_]*|reduce_kernel|CatArray\w*|copyBuffer|fillBuffer\w*|scan_impl|init_lookback\w*|transform_impl|radix\w*|merge\w*)", name)
    short = (m.group(1) if m else name)[:44]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
    dur = (e - s) / 1e3
    print("%9.1f  +%6.1f gap  %7.1f us  %s" % ((s - t0) / 1e3, gap, dur, short))
    tot_k += dur
    tot_gap += max(gap, 0.0)
    prev_end = max(e, prev_end or e)
    a = agg.setdefault(short, [0, 0.0])
    a[0] += 1
    a[1] += dur
print("kernels %d  kernel time %.1f us  gaps %.1f us  span %.1f us" %
      (len(step), tot_k, tot_gap, (prev_end - t0) / 1e3))
