#!/bin/bash
# GEMM tests + grad parity + default bench on the 2-chain / pinned-conversion kernel
mkdir -p gpurun_out; export TMPDIR=/tmp
S=gpurun_out/r05_visit_l_summary.txt; : > $S
timeout 1500 python -m pytest tests/test_gpu_gemm_x6.py tests/test_gpu_grad_parity.py tests/test_gpu_kernels.py -m gpu -q --timeout 900 -p no:cacheprovider -k "gemm or x6 or head or grad" > gpurun_out/pytest_l.log 2>&1
echo "pytest exit $?" | tee -a $S
grep -E "^(FAILED|ERROR)|passed|failed|x6 a/b" gpurun_out/pytest_l.log | tail -30 | tee -a $S
grep -E "^E  " gpurun_out/pytest_l.log | head -20 | cut -c1-300 | tee -a $S
timeout 900 python bench.py --steps 50 --warmup 20 --no-cpu-baseline --no-parity --no-uniform > gpurun_out/bench_l.json 2> gpurun_out/bench_l.err
echo "bench exit $?" | tee -a $S
python - gpurun_out/bench_l.json <<'PY' | tee -a $S
import json, sys
try:
    d = json.load(open(sys.argv[1]))
except Exception as e:
    print("no bench json:", e); sys.exit(0)
def show(tag, x):
    r = x.get("roofline", {})
    print("%s: %.0f samples/s  %.4f ms/step (young %s)  gemm %.1f us/step frac %.3f  sparse %.1f us" % (tag, x["value"], x["ms_per_step"], x.get("young_run", {}).get("ms_per_step"), r.get("gemm_us_per_step", 0), r.get("frac", 0), x.get("roofline_sparse", {}).get("us_per_step", 0)))
    for k, v in sorted(r.get("by_shape_MxNxK", {}).items()):
        print("    %-40s x%.0f  %7.2f us  %6.1f TF  %.3f" % (k[:40], v["launches_per_step"], v["avg_launch_us"], v["tflops"], v["frac"]))
show("deepfm", d)
if "dcnv2" in d: show("dcnv2", d["dcnv2"])
print({k: d.get(k) for k in ("warmup", "warmup_requested", "dtype", "kernel_sum_us", "wall_minus_kernel_sum_us")})
PY
tail -2 gpurun_out/bench_l.err | cut -c1-300 | tee -a $S
