#!/usr/bin/env python3
"""SQ / GRBM counters of the GEMM lab's kernels (scripts/gpu_r03a.sh): per kernel x grid the matrix-pipe
busy fraction, what the waves wait on, and the EFFECTIVE clock (GRBM_GUI_ACTIVE / duration).
usage: pmc_lab.py counter_collection.csv kernel_trace.csv"""
import csv
import re
import sys

dur = {}
for r in csv.DictReader(open(sys.argv[2])):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
agg = {}
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(k_gemm_f32_pipe<[^>]*>|k_gemm_f32_pair<[^>]*>|k_splitk_reduce\w*)", r["Kernel_Name"])
    if not m:
        continue
    key = (m.group(1), r.get("Grid_Size", ""))
    a = agg.setdefault(key, {"ids": set()})
    a[r["Counter_Name"]] = a.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    a["ids"].add(r["Dispatch_Id"])
for key, a in sorted(agg.items()):
    n = len(a["ids"])
    us = sum(dur.get(i, 0.0) for i in a["ids"]) / max(n, 1)
    g = lambda k: a.get(k, 0.0) / max(n, 1)
    wc = max(g("SQ_WAVE_CYCLES"), 1.0)
    clk = g("GRBM_GUI_ACTIVE") / max(us * 1e3, 1e-9)
    # SQ_VALU_MFMA_BUSY_CYCLES: cycles summed over the SIMDs that report (per-XCD sampling: see the
    # ratio to duration x clock x 1024 SIMDs only as a relative figure between kernels)
    print("%-58s grid %-9s x%-3d %7.1f us | clk %.2f GHz | MFMA busy %.3e (%.1f/ns) | waves: parked %.0f%% "
          "issue-stall %.0f%% active %.0f%% | LDS conflict %.1f%%"
          % (key[0][:58], key[1], n, us, clk, g("SQ_VALU_MFMA_BUSY_CYCLES"),
             g("SQ_VALU_MFMA_BUSY_CYCLES") / max(us * 1e3, 1), 100 * g("SQ_WAIT_ANY") / wc,
             100 * g("SQ_WAIT_INST_ANY") / wc, 100 * g("SQ_ACTIVE_INST_ANY") / wc,
             100 * g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_LDS_IDX_ACTIVE"), 1)))
