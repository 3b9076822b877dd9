#!/bin/bash
# round 5 visit e: the split-bf16 GEMM in the product — GEMM tests, then the default bench with it on and off
mkdir -p gpurun_out; export TMPDIR=/tmp
S=gpurun_out/r05_visit_e_summary.txt; : > $S
timeout 1200 python -m pytest tests/test_gpu_gemm_x6.py tests/test_gpu_gemm_edge_bodies.py "tests/test_gpu_kernels.py" -m gpu -q -x --timeout 900 -p no:cacheprovider -k "gemm or x6 or head or body or slab" > gpurun_out/pytest_e.log 2>&1
echo "pytest exit $?" | tee -a $S
grep -E "^(FAILED|ERROR)|passed|failed|x6 a/b" gpurun_out/pytest_e.log | tail -30 | tee -a $S
grep -E "^E  " gpurun_out/pytest_e.log | head -20 | cut -c1-300 | tee -a $S
for x in 1 0; do
  FX_GEMM_BF16X6=$x timeout 600 python bench.py --steps 50 --warmup 20 --no-cpu-baseline > gpurun_out/bench_e_x6_$x.json 2> gpurun_out/bench_e_x6_$x.err
  echo "bench x6=$x exit $?" | tee -a $S
  python - gpurun_out/bench_e_x6_$x.json <<'PY' | tee -a $S
import json, sys
try:
    d = json.load(open(sys.argv[1]))
except Exception as e:
    print("no bench json:", e); sys.exit(0)
def show(tag, x):
    r = x.get("roofline", {})
    print("%s: %.0f samples/s  %.4f ms/step  gemm %.1f us/step frac %.3f" % (tag, x["value"], x["ms_per_step"], r.get("gemm_us_per_step", 0), r.get("frac", 0)))
    for k, v in sorted(r.get("by_shape_MxNxK", {}).items()):
        print("    %-40s x%.0f  %7.2f us  %6.1f TF  %.3f" % (k, v["launches_per_step"], v["avg_launch_us"], v["tflops"], v["frac"]))
show("deepfm", d)
if "dcnv2" in d: show("dcnv2", d["dcnv2"])
PY
  tail -2 gpurun_out/bench_e_x6_$x.err | cut -c1-300 | tee -a $S
done
