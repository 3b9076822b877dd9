#!/bin/bash
# HBM-side traffic of the GEMM kernels inside the DeepFM step, per launch: rocprofv3 --pmc in SEPARATE
# passes (FETCH_SIZE, WRITE_SIZE), kernel-trace only (MI355X_MICROARCH.md, HBM section).
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$C -- \
      python $REPO/bench.py --steps 4 --warmup 5 --no-graph --no-cpu-baseline --no-kernel-timing --no-dcnv2 > /dev/null 2> $OUT/pmc_$C.err)
  F=$(find /tmp/pmc_$C -name '*counter_collection.csv' | head -1)
  python - "$F" "$C" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = {}
for r in rows:
    if r.get("Counter_Name") != sys.argv[2]:
        continue
    m = re.search(r"(k_gemm_f32_pipe<[^>]*>|k_gemm_f32_pair<[^>]*>|k_emb_fm_fwd<\d>|k_emb_gather_fwd<\d>|k_mt_adam|k_sparse_update_multi<\w+>|k_sparse_adam<4>)", r["Kernel_Name"])
    if not m:
        continue
    a = agg.setdefault(m.group(1), [0, 0.0])
    a[0] += 1
    a[1] += float(r["Counter_Value"])
for k, (n, v) in sorted(agg.items()):
    print("%-11s %-44s launches %4d  avg per launch %12.1f (counter units = KB)" % (sys.argv[2], k, n, v / n))
PY
done
