#!/bin/bash
# round 5 visit c: PMC counters of the in-kernel-split kernel on 4096^3 (what do the waves wait for?)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r05_gemm_x6s_pmc.txt
: > $O
cd /tmp
rocprofv3 -L > $R/gpurun_out/r05_rocprof_counters.txt 2>&1
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVES SQ_INSTS_WAVE32_LDS"; do
  rm -rf /tmp/pmc
  timeout 120 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc --output-format csv -- $R/scripts/ubench/gemm_x6s_lab nolib prof > /tmp/pmc.log 2>&1
  echo "== $set (rc $?)" >> $O
  f=$(find /tmp/pmc -name '*counter_collection.csv' | head -1)
  python3 - "$f" >> $O <<'PY'
import csv, sys, collections
try:
    rows = list(csv.DictReader(open(sys.argv[1])))
except Exception as e:
    print("no csv", e); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "n=%d" % len(next(iter(d.values()))))
PY
  tail -3 /tmp/pmc.log >> $O
done
cat $O | cut -c1-600
