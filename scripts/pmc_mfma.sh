#!/bin/bash
# SQ counters (one --pmc pass, kernel-trace only) of the MFMA kernels inside the DCNv2 and DIN steps (MODELS=... for others):
# matrix-pipe busy cycles per kernel next to its duration (eager launches, so every dispatch is attributed).
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
CTRS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
for M in ${MODELS:-DCNv2 DIN}; do
  rm -rf /tmp/pmc_mfma_$M
  (cd /tmp && timeout 300 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d /tmp/pmc_mfma_$M -- \
      python $REPO/bench.py --model $M --steps 3 --warmup 5 --no-graph --no-cpu-baseline --no-kernel-timing > /dev/null 2> $OUT/pmc_mfma_$M.err)
  F=$(find /tmp/pmc_mfma_$M -name '*counter_collection.csv' | head -1)
  T=$(find /tmp/pmc_mfma_$M -name '*kernel_trace.csv' | head -1)
  echo "== $M"
  python - "$F" "$T" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
dur = {}
for r in csv.DictReader(open(sys.argv[2])):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
agg = {}
for r in rows:
    name = r["Kernel_Name"]
    m = re.search(r"(k_gemm_f32_pipe<[^>]*>|k_gemm_f32_pair<[^>]*>|k_din_attn2?_\w+<[^>]*>|k_dot_interact_\w+|k_cin_\w+<[^>]*>|k_gemm_f32_multi)", name)
    if not m:
        continue
    key = (m.group(1), r.get("Grid_Size", ""))
    a = agg.setdefault(key, {"ids": set()})
    c = a.setdefault(r["Counter_Name"], 0.0)
    a[r["Counter_Name"]] = c + float(r["Counter_Value"])
    a["ids"].add(r["Dispatch_Id"])
for key, a in sorted(agg.items()):
    n = len(a["ids"])
    us = sum(dur.get(i, 0.0) for i in a["ids"]) / max(n, 1)
    g = lambda k: a.get(k, 0.0) / max(n, 1)
    wc = max(g("SQ_WAVE_CYCLES"), 1.0)
    print("%-62s grid %-9s x%-3d %7.1f us | MFMA busy %.3e cyc (%.1f cyc/ns) | busy %.3e | waves: parked %.0f%% issue-stall %.0f%% active %.0f%% | LDS conflict %.1f%%"
          % (key[0][:62], key[1], n, us, g("SQ_VALU_MFMA_BUSY_CYCLES"), g("SQ_VALU_MFMA_BUSY_CYCLES") / max(us * 1e3, 1),
             g("SQ_BUSY_CYCLES"), 100 * g("SQ_WAIT_ANY") / wc, 100 * g("SQ_WAIT_INST_ANY") / wc,
             100 * g("SQ_ACTIVE_INST_ANY") / wc, 100 * g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_LDS_IDX_ACTIVE"), 1)))
PY
done
