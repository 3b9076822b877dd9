#!/bin/bash
# Round-6 GPU visit, parameterised (one script for the round): tests, A/B of environment switches, timeline.
# usage: bash scripts/gpu_r06.sh TAG "pytest args or empty" "ENV_A|ENV_B or empty" MODEL [reps] [timeline: 0/1]
TAG=$1; PYT=$2; AB=$3; MODEL=${4:-DeepFM}; REPS=${5:-2}; TL=${6:-1}
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
S=$OUT/summary_$TAG.txt; : > $S
if [ -n "$PYT" ]; then
  echo "== pytest $PYT" | tee -a $S
  timeout 2400 python -m pytest $PYT -m gpu -q --timeout 900 -p no:cacheprovider -s > $OUT/pytest_$TAG.log 2>&1
  echo "pytest exit $?" | tee -a $S
  grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_$TAG.log | tail -25 | tee -a $S
  grep -E "^E  " $OUT/pytest_$TAG.log | head -30 | cut -c1-400 | tee -a $S
  grep -E "^\[(series|quad|exact|x6 range|long horizon|baseline)" $OUT/pytest_$TAG.log | head -60 | tee -a $S
fi
if [ -n "$AB" ]; then
  A="${AB%%|*}"; B="${AB##*|}"
  for R in $(seq 1 $REPS); do
    for V in A B; do
      if [ $V = A ]; then E="$A"; else E="$B"; fi
      env $E timeout 400 python bench.py --model $MODEL --steps 100 --warmup 10 --no-cpu-baseline --no-dcnv2 --no-din --no-parity --no-uniform 2>$OUT/ab_$TAG.err | head -1 > $OUT/ab_tmp.json
      python - "$V [$E]" $OUT/ab_tmp.json <<'PY' | tee -a $S
import json, sys
try:
    d = json.load(open(sys.argv[2]))
except Exception as e:
    print(sys.argv[1], "no json", e); sys.exit(0)
sp = d.get("roofline_sparse") or {}
y = d.get("young_run") or {}
print(sys.argv[1], round(d["value"]), "ms %.4f" % d["ms_per_step"], "young %.4f" % y.get("ms_per_step", 0),
      "sparse %.1f us frac %.3f" % (sp.get("us_per_step", 0), sp.get("frac", 0)), "gemm frac %.3f" % d.get("roofline", {}).get("frac", 0))
PY
    done
  done
  tail -3 $OUT/ab_$TAG.err | tee -a $S
fi
if [ "$TL" = "1" ]; then
  echo "== rocprofv3 kernel trace ($MODEL)" | tee -a $S
  rm -rf /tmp/prof_$TAG
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- \
      python $REPO/bench.py --model $MODEL --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-dcnv2 --no-din --no-parity --no-uniform > $OUT/prof_bench_$TAG.json 2> $OUT/prof_$TAG.err)
  echo "rocprof exit $?" | tee -a $S
  STATS=$(ls -t $(find /tmp/prof_$TAG -name '*kernel_stats.csv') 2>/dev/null | head -1)
  if [ -n "$STATS" ]; then cp $STATS $OUT/kernel_stats_$TAG.csv; fi
  TR=$(find /tmp/prof_$TAG -name '*kernel_trace.csv' | head -1)
  python scripts/step_timeline.py $TR 3 > $OUT/timeline_$TAG.txt; cat $OUT/timeline_$TAG.txt | tail -40 | tee -a $S
fi
