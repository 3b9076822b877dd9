#!/bin/bash
# quad catch-up: tests, then the default bench with it on and off (steady state is what it is for)
mkdir -p gpurun_out; export TMPDIR=/tmp
S=gpurun_out/r05_visit_n_summary.txt; : > $S
timeout 1500 python -m pytest tests/test_gpu_fused.py tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_full_vocab_parity.py tests/test_gpu_bf16.py -m gpu -q -s --timeout 900 -p no:cacheprovider -k "catchup or adam or deepfm or full or bf16 or golden or fused" > gpurun_out/pytest_n.log 2>&1
echo "pytest exit $?" | tee -a $S
grep -E "^(FAILED|ERROR)|passed|failed|quad catch-up" gpurun_out/pytest_n.log | tail -30 | tee -a $S
grep -E "^E  " gpurun_out/pytest_n.log | head -20 | cut -c1-300 | tee -a $S
for q in 1 0; do
  FX_CATCHUP_QUAD=$q timeout 900 python bench.py --steps 50 --warmup 20 --no-cpu-baseline --no-parity --no-uniform --no-dcnv2 > gpurun_out/bench_n_$q.json 2> gpurun_out/bench_n_$q.err
  python - gpurun_out/bench_n_$q.json $q <<'PY' | tee -a $S
import json, sys
d = json.load(open(sys.argv[1]))
sp = d.get("roofline_sparse", {})
print("FX_CATCHUP_QUAD=%s: steady %.4f ms/step (%.0f samples/s), young %.4f, sparse path %.1f us, kernel sum %.1f" % (sys.argv[2], d["ms_per_step"], d["value"], d["young_run"]["ms_per_step"], sp.get("us_per_step", 0), d.get("kernel_sum_us", 0)))
PY
done
