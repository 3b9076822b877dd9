#!/bin/bash
# One GPU-box visit (round 2): chosen test files, bench line, rocprofv3 kernel trace + step timeline.
# usage: bash scripts/gpu_visit.sh TAG "pytest args" [bench args...]
TAG=$1; shift
PYT=$1; shift
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
S=$OUT/summary_$TAG.txt
: > $S
if [ -n "$PYT" ]; then
  echo "== pytest $PYT" | tee -a $S
  timeout 1500 python -m pytest $PYT -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/pytest_$TAG.log 2>&1
  echo "pytest exit $?" | tee -a $S
  grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_$TAG.log | tail -25 | tee -a $S
  grep -E "^E  " $OUT/pytest_$TAG.log | head -30 | cut -c1-400 | tee -a $S
fi
echo "== bench $@" | tee -a $S
timeout 900 python bench.py --steps 50 --warmup 10 "$@" > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
echo "bench exit $?" | tee -a $S
python - $OUT/bench_$TAG.json <<'PY' | tee -a $S
import json, sys
try:
    d = json.load(open(sys.argv[1]))
except Exception as e:
    print("no bench json:", e); sys.exit(0)
def show(tag, x):
    r = x.get("roofline", {})
    print("%s: %.0f samples/s  %.4f ms/step  gemm frac %.3f (%.1f us/step, %d launches)" % (
        tag, x["value"], x["ms_per_step"], r.get("frac", 0), r.get("gemm_us_per_step", 0), r.get("launches", 0)))
    for k, v in sorted(r.get("by_shape_MxNxK", {}).items()):
        print("    %-18s x%.0f  %7.2f us  %6.1f TF  %.3f" % (k, v["launches_per_step"], v["avg_launch_us"], v["tflops"], v["frac"]))
    sp = x.get("roofline_sparse")
    if sp: print("    sparse path: %.1f us/step, %d launches, %.0f GB/s (%.3f of HBM)" % (sp["us_per_step"], sp["launches_per_step"], sp["achieved"], sp["frac"]))
show("main", d)
if "dcnv2" in d: show("dcnv2", d["dcnv2"])
if "cpu_baseline" in d: print("cpu:", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
tail -3 $OUT/bench_$TAG.err | tee -a $S
echo "== rocprofv3 kernel trace" | tee -a $S
rm -rf /tmp/prof_$TAG
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- \
    python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-dcnv2 "$@" > $OUT/prof_bench_$TAG.json 2> $OUT/prof_$TAG.err)
echo "rocprof exit $?" | tee -a $S
STATS=$(ls -t $(find /tmp/prof_$TAG -name '*kernel_stats.csv') 2>/dev/null | head -1)
if [ -n "$STATS" ]; then cp $STATS $OUT/kernel_stats_$TAG.csv; fi
TR=$(find /tmp/prof_$TAG -name '*kernel_trace.csv' | head -1)
python scripts/step_timeline.py $TR 3 > $OUT/timeline_$TAG.txt; tail -1 $OUT/timeline_$TAG.txt | tee -a $S
