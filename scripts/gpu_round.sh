#!/bin/bash
# One GPU-box visit: smoke, GPU parity tests, bench line, rocprofv3 kernel trace.
# Usage (from the repo root on the GPU box):  bash scripts/gpu_round.sh [tag] [steps]
TAG=${1:-r01}
STEPS=${2:-30}
REPO=$PWD
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
echo "== smoke" | tee $OUT/summary_$TAG.txt
timeout 600 python __graft_entry__.py --smoke > $OUT/smoke_$TAG.log 2>&1; echo "smoke exit $?" | tee -a $OUT/summary_$TAG.txt
tail -3 $OUT/smoke_$TAG.log | tee -a $OUT/summary_$TAG.txt
echo "== pytest -m gpu" | tee -a $OUT/summary_$TAG.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $OUT/pytest_gpu_$TAG.log 2>&1
echo "pytest exit $?" | tee -a $OUT/summary_$TAG.txt
tail -40 $OUT/pytest_gpu_$TAG.log | tee -a $OUT/summary_$TAG.txt
echo "== bench" | tee -a $OUT/summary_$TAG.txt
timeout 900 python bench.py --steps $STEPS --warmup 10 > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
echo "bench exit $?" | tee -a $OUT/summary_$TAG.txt
cat $OUT/bench_$TAG.json | tee -a $OUT/summary_$TAG.txt
tail -5 $OUT/bench_$TAG.err | tee -a $OUT/summary_$TAG.txt
echo "== rocprofv3 kernel trace" | tee -a $OUT/summary_$TAG.txt
rm -rf $OUT/prof_$TAG
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- \
    python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > $OUT/prof_bench_$TAG.json 2> $OUT/prof_$TAG.err)
echo "rocprof exit $?" | tee -a $OUT/summary_$TAG.txt
cat $OUT/prof_bench_$TAG.json | tee -a $OUT/summary_$TAG.txt
STATS=$(ls -t $(find $OUT/prof_$TAG -name '*kernel_stats.csv') 2>/dev/null | head -1)   # newest: merged-back dirs can hold older runs
if [ -n "$STATS" ]; then cp $STATS $OUT/kernel_stats_$TAG.csv; head -40 $STATS | tee -a $OUT/summary_$TAG.txt; fi
# keep the merged-back payload small
find $OUT/prof_$TAG -name '*kernel_trace.csv' -size +20M -delete
