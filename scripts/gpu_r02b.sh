#!/bin/bash
# round 2, visit B: smoke, the whole GPU suite, bench line (DeepFM + DCNv2 sub-object), rocprofv3
# kernel stats of the same command, one-step timelines of DeepFM / DCNv2 / DIN.
TAG=${1:-r02b}
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
S=$OUT/summary_$TAG.txt
echo "== smoke" | tee $S
timeout 600 python __graft_entry__.py --smoke > $OUT/smoke_$TAG.log 2>&1; echo "smoke exit $?" | tee -a $S
tail -2 $OUT/smoke_$TAG.log | tee -a $S
echo "== pytest -m gpu" | tee -a $S
FX_PARITY_REPORT=$OUT/parity_$TAG.jsonl timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider --durations=15 > $OUT/pytest_gpu_$TAG.log 2>&1
echo "pytest exit $?" | tee -a $S
tail -30 $OUT/pytest_gpu_$TAG.log | tee -a $S
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
if [ -n "$STATS" ]; then cp $STATS $OUT/kernel_stats_deepfm_$TAG.csv; python scripts/kstats.py $STATS 20 30 | tee -a $S; fi
TR=$(find /tmp/prof_$TAG -name '*kernel_trace.csv' | head -1)
python scripts/step_timeline.py $TR 3 > $OUT/timeline_deepfm_$TAG.txt; tail -1 $OUT/timeline_deepfm_$TAG.txt | tee -a $S
for M in DCNv2 DIN; do
  echo "== timeline $M" | tee -a $S
  rm -rf /tmp/prof_${TAG}_$M
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_$M -- \
      python $REPO/bench.py --model $M --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > $OUT/prof_bench_${TAG}_$M.json 2> $OUT/prof_${TAG}_$M.err)
  cat $OUT/prof_bench_${TAG}_$M.json | cut -c1-300 | tee -a $S
  STATS=$(ls -t $(find /tmp/prof_${TAG}_$M -name '*kernel_stats.csv') 2>/dev/null | head -1)
  if [ -n "$STATS" ]; then cp $STATS $OUT/kernel_stats_${M}_$TAG.csv; fi
  TR=$(find /tmp/prof_${TAG}_$M -name '*kernel_trace.csv' | head -1)
  python scripts/step_timeline.py $TR 3 > $OUT/timeline_${M}_$TAG.txt; tail -1 $OUT/timeline_${M}_$TAG.txt | tee -a $S
done
