#!/bin/bash
# SQ counters of fx_gemm_f32 vs the rocBLAS/hipBLASLt fp32 kernel torch.mm picks, on the tower shapes
# (scripts/gemm_vs_blas.py): one rocprofv3 --pmc pass (8 SQ slots), kernel-trace only.
# Keep it to these 8: a pass with GRBM_GUI_ACTIVE + SQ_WAVES added (10 counters) never returned on the GPU box.
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
CTRS="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
rm -rf /tmp/pmc_gemm
(cd /tmp && timeout 300 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d /tmp/pmc_gemm -- \
    python $REPO/scripts/gemm_vs_blas.py > $OUT/pmc_gemm.out 2> $OUT/pmc_gemm.err)
echo "exit $?"
F=$(find /tmp/pmc_gemm -name '*counter_collection.csv' | head -1)
python - "$F" <<'PY' | tee $OUT/pmc_gemm_summary.txt
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = {}
for r in rows:
    name = r["Kernel_Name"]
    if not ("k_gemm_f32" in name or name.startswith("Cijk")):
        continue
    m = re.search(r"(k_gemm_f32_pipe<[^>]*>|k_gemm_f32<[^>]*>|Cijk[A-Za-z0-9_]*?MT\d+x\d+x\d+)", name)
    short = m.group(1) if m else name[:40]
    key = (short, r.get("Grid_Size", ""), r.get("LDS_Block_Size", ""))
    a = agg.setdefault(key, {})
    c = a.setdefault(r["Counter_Name"], [0, 0.0])
    c[0] += 1
    c[1] += float(r["Counter_Value"])
for key, a in sorted(agg.items()):
    n = a["SQ_WAVE_CYCLES"][0]
    g = lambda k: a.get(k, [1, 0.0])[1] / max(a.get(k, [1, 0.0])[0], 1)
    wc = g("SQ_WAVE_CYCLES")
    print("%s grid=%s lds=%s launches=%d" % (key[0][-60:], key[1], key[2], n))
    print("   wave_cycles %.3e | parked(WAIT_ANY) %.1f%% | issue-stall(WAIT_INST_ANY) %.1f%% | active %.1f%% | "
          "lds-issue-stall %.1f%% | MFMA busy cyc %.3e | LDS conflict/active %.1f%%" %
          (wc, 100 * g("SQ_WAIT_ANY") / wc, 100 * g("SQ_WAIT_INST_ANY") / wc, 100 * g("SQ_ACTIVE_INST_ANY") / wc,
           100 * g("SQ_WAIT_INST_LDS") / wc, g("SQ_VALU_MFMA_BUSY_CYCLES"),
           100 * g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_LDS_IDX_ACTIVE"), 1)))
PY
tail -12 $OUT/pmc_gemm.out
