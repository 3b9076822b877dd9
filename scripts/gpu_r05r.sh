#!/bin/bash
# effective clock + MFMA busy of the GEMM kernels (lab uniform / lab specialised / production), 4096^3 and the step's shape
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r05_gemm_clock_pmc.txt; : > $O
cd /tmp
for set in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA"; do
  rm -rf /tmp/pmc
  FX_GEMM_BF16X6=1 timeout 200 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc --output-format csv -- $R/scripts/ubench/gemm_x6w_lab $R/fuxictr_amd/libfxctr.so quick > /tmp/pmc.log 2>&1
  python3 - $(find /tmp/pmc -name '*counter_collection.csv' | head -1) $(find /tmp/pmc -name '*kernel_trace.csv' | head -1) >> $O <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
tr = list(csv.DictReader(open(sys.argv[2])))
dur = collections.defaultdict(list)
for r in tr:
    dur[(r["Kernel_Name"][:44], r.get("Grid_Size_X", r.get("Grid_Size", "")))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    acc[(r["Kernel_Name"][:44], r.get("Grid_Size_X", r.get("Grid_Size", "")))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    if "gemm" not in k[0]: continue
    c = {n: sum(v) / len(v) for n, v in d.items()}
    us = sorted(dur[k])[len(dur[k]) // 2] if k in dur else 0
    gui = c.get("GRBM_GUI_ACTIVE", 0)
    print("%-46s grid %-7s %8.1f us  GUI_ACTIVE %.0f -> %.2f GHz | MFMA busy %.0f (%.1f%% of 1024 SIMDs x GUI) | wave cyc x4 %.3g wait_any %.0f%% wait_inst %.0f%% active %.0f%%" % (
        k[0], k[1], us, gui, gui / us / 1e3 if us else 0, c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0),
        100 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / gui if gui else 0, 4 * c.get("SQ_WAVE_CYCLES", 0),
        100 * c.get("SQ_WAIT_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1), 100 * c.get("SQ_WAIT_INST_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1),
        100 * c.get("SQ_ACTIVE_INST_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1)))
PY
done
cat $O
