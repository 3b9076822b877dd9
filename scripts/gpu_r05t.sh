#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
S=gpurun_out/r05_visit_t_summary.txt; : > $S
timeout 600 python -m pytest tests/test_gpu_gemm_x6.py tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "x6 or gemm" 2>&1 | tail -3 | tee -a $S
for q in 1 0; do
  FX_X6_PLAN=$q timeout 900 python bench.py --steps 50 --warmup 20 --age-steps 60 --no-cpu-baseline --no-parity --no-uniform > gpurun_out/bench_t_$q.json 2> gpurun_out/bench_t_$q.err
  python - gpurun_out/bench_t_$q.json $q <<'PY' | tee -a $S
import json, sys
d = json.load(open(sys.argv[1]))
for tag, x in (("deepfm", d), ("dcnv2", d["dcnv2"])):
    r = x["roofline"]
    print("FX_X6_PLAN=%s %s: %.4f ms/step  gemm %.1f us/step" % (sys.argv[2], tag, x["ms_per_step"], r["gemm_us_per_step"]))
    for k, v in sorted(r.get("by_shape_MxNxK", {}).items()):
        print("    %-60s x%.0f  %7.2f us" % (k[:60], v["launches_per_step"], v["avg_launch_us"]))
PY
done
