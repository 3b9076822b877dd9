#!/bin/bash
# round 2, visit C: fused DIN attention — kernel tests, DIN model / baseline-shape parity, bench + timeline
TAG=${1:-r02c}
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
S=$OUT/summary_$TAG.txt
echo "== pytest fused DIN attention" | tee $S
timeout 900 python -m pytest tests/test_gpu_din_attn.py -m gpu -q --timeout 600 -p no:cacheprovider > $OUT/pytest_dinattn_$TAG.log 2>&1
echo "pytest exit $?" | tee -a $S
tail -25 $OUT/pytest_dinattn_$TAG.log | tee -a $S
echo "== pytest DIN models" | tee -a $S
FX_PARITY_REPORT=$OUT/parity_$TAG.jsonl timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_baseline_shapes.py tests/test_gpu_dist.py -m gpu -q -k "din or DIN" --timeout 600 -p no:cacheprovider > $OUT/pytest_din_$TAG.log 2>&1
echo "pytest exit $?" | tee -a $S
tail -15 $OUT/pytest_din_$TAG.log | tee -a $S
for V in 1 0; do
  echo "== bench DIN (FX_DIN_FUSED=$V)" | tee -a $S
  FX_DIN_FUSED=$V timeout 600 python bench.py --model DIN --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench_din_fused${V}_$TAG.json 2> $OUT/bench_din_fused${V}_$TAG.err
  echo "bench exit $?" | tee -a $S
  cut -c1-400 $OUT/bench_din_fused${V}_$TAG.json | tee -a $S
  tail -3 $OUT/bench_din_fused${V}_$TAG.err | tee -a $S
done
echo "== timeline DIN" | tee -a $S
rm -rf /tmp/prof_$TAG
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- \
    python $REPO/bench.py --model DIN --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > $OUT/prof_bench_$TAG.json 2> $OUT/prof_$TAG.err)
STATS=$(ls -t $(find /tmp/prof_$TAG -name '*kernel_stats.csv') 2>/dev/null | head -1)
if [ -n "$STATS" ]; then cp $STATS $OUT/kernel_stats_din_$TAG.csv; fi
TR=$(find /tmp/prof_$TAG -name '*kernel_trace.csv' | head -1)
python scripts/step_timeline.py $TR 3 > $OUT/timeline_din_$TAG.txt; cat $OUT/timeline_din_$TAG.txt | cut -c1-110 | tee -a $S
