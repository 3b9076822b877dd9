#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
S=gpurun_out/r05_visit_o_summary.txt; : > $S
timeout 1500 python -m pytest tests/test_gpu_fused.py tests/test_gpu_models.py tests/test_gpu_baseline_shapes.py tests/test_gpu_full_vocab_parity.py -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/pytest_o.log 2>&1
echo "pytest exit $?" | tee -a $S
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest_o.log | tail -30 | tee -a $S
grep -E "^E  " gpurun_out/pytest_o.log | head -20 | cut -c1-300 | tee -a $S
for q in 1 0; do
  FX_CATCHUP_QUAD=$q timeout 900 python bench.py --steps 50 --warmup 20 --no-cpu-baseline --no-parity --no-uniform > gpurun_out/bench_o_$q.json 2> gpurun_out/bench_o_$q.err
  python - gpurun_out/bench_o_$q.json $q <<'PY' | tee -a $S
import json, sys
d = json.load(open(sys.argv[1]))
for tag, x in (("deepfm", d), ("dcnv2", d["dcnv2"])):
    sp = x.get("roofline_sparse", {})
    print("FX_CATCHUP_QUAD=%s %s: steady %.4f ms/step (%.0f samples/s), young %.4f, sparse path %.1f us, kernel sum %.1f" % (sys.argv[2], tag, x["ms_per_step"], x["value"], x["young_run"]["ms_per_step"], sp.get("us_per_step", 0), x.get("kernel_sum_us", 0)))
PY
done
