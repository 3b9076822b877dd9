#!/bin/bash
# kernel-trace durations: lab kernel vs production kernel, same process
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
cd /tmp && rm -rf /tmp/kt && FX_GEMM_BF16X6=1 timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/kt --output-format csv -- $R/scripts/ubench/gemm_x6s_lab $R/fuxictr_amd/libfxctr.so quick > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name '*kernel_stats.csv' | head -1)
cp $f $R/gpurun_out/r05_gemm_lab_vs_prod_kernel_stats.csv
python3 - $(find /tmp/kt -name '*kernel_trace.csv' | head -1) <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(list)
for r in rows:
    d[(r["Kernel_Name"][:50], r.get("Grid_Size_X", r.get("Grid_Size", "")))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    v2 = sorted(v)
    print("%-52s grid %-8s n=%3d  min %8.2f  median %8.2f  max %8.2f us" % (k[0], k[1], len(v), v2[0], v2[len(v2)//2], v2[-1]))
PY
