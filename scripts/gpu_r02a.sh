#!/bin/bash
# round 2, visit A: baseline-shape parity tests, bench line (DeepFM + DCNv2 sub-object), rocprof trace
TAG=${1:-r02a}
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
S=$OUT/summary_$TAG.txt
echo "== baseline-shape parity" | tee $S
FX_PARITY_REPORT=$OUT/parity_$TAG.jsonl timeout 1200 python -m pytest tests/test_gpu_baseline_shapes.py -m gpu -q -x --timeout 900 -p no:cacheprovider -s > $OUT/pytest_bs_$TAG.log 2>&1
echo "pytest exit $?" | tee -a $S
grep -E "baseline-shape parity|passed|failed|Error" $OUT/pytest_bs_$TAG.log | tail -20 | tee -a $S
echo "== bench" | tee -a $S
timeout 900 python bench.py --steps 50 --warmup 10 > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
echo "bench exit $?" | tee -a $S
cat $OUT/bench_$TAG.json | tee -a $S
tail -5 $OUT/bench_$TAG.err | tee -a $S
echo "== rocprofv3 kernel trace (DeepFM, graph replay)" | tee -a $S
rm -rf /tmp/prof_$TAG
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- \
    python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-dcnv2 > $OUT/prof_bench_$TAG.json 2> $OUT/prof_$TAG.err)
echo "rocprof exit $?" | tee -a $S
STATS=$(ls -t $(find /tmp/prof_$TAG -name '*kernel_stats.csv') 2>/dev/null | head -1)
if [ -n "$STATS" ]; then cp $STATS $OUT/kernel_stats_$TAG.csv; python scripts/kstats.py $STATS 20 30 | tee -a $S; fi
TR=$(find /tmp/prof_$TAG -name '*kernel_trace.csv' | head -1)
python scripts/step_timeline.py $TR 3 > $OUT/timeline_$TAG.txt; tail -1 $OUT/timeline_$TAG.txt | tee -a $S
