#!/bin/bash
# the driver's bench command as is (all legs) + the full-vocabulary parity test
mkdir -p gpurun_out; export TMPDIR=/tmp
S=gpurun_out/r05_visit_m_summary.txt; : > $S
T0=$(date +%s); timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_m.json 2> gpurun_out/bench_m.err
echo "bench exit $?" | tee -a $S
echo "bench wall $(( $(date +%s) - T0 )) s" | tee -a $S
python - gpurun_out/bench_m.json <<'PY' | tee -a $S
import json, sys
d = json.load(open(sys.argv[1]))
for k in ("value", "ms_per_step", "warmup", "warmup_requested", "dtype", "young_run", "value_uniform", "uniform", "parity_full_vocab", "cpu_baseline"):
    print(k, "=", json.dumps(d.get(k))[:900])
print("dcnv2:", d["dcnv2"]["value"], d["dcnv2"]["ms_per_step"], d["dcnv2"].get("young_run"))
print("roofline:", {k: v for k, v in d["roofline"].items() if k in ("achieved", "peak", "frac", "peak_note")})
print("sparse:", {k: v for k, v in d["roofline_sparse"].items() if k in ("us_per_step", "frac", "per_kernel")})
PY
timeout 1200 python -m pytest tests/test_gpu_gemm_x6.py tests/test_gpu_kernels.py -k "gemm or x6" -m gpu -q -s --timeout 900 -p no:cacheprovider > gpurun_out/pytest_m.log 2>&1
echo "pytest exit $?" | tee -a $S
grep -E "^(FAILED|ERROR)|passed|failed|full-vocab" gpurun_out/pytest_m.log | cut -c1-1200 | tee -a $S
grep -E "^E  " gpurun_out/pytest_m.log | head -20 | cut -c1-400 | tee -a $S
