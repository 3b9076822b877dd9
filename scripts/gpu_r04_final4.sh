#!/bin/bash
# round 4, closing check of the catch-up replay in windows of 8 steps: the tests that reach it, default bench line,
# steady state, DeepFM kernel stats + timeline.
TAG=${1:-r04final4}
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
S=$OUT/summary_$TAG.txt
echo "== tests that reach the catch-up replay (kernels, models, fused, shard kernels, bf16, baseline shapes)" | tee $S
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_fused.py tests/test_gpu_shard_kernels.py tests/test_gpu_bf16.py tests/test_gpu_baseline_shapes.py -m gpu -q --timeout 600 -p no:cacheprovider > $OUT/pytest_gpu_$TAG.log 2>&1
echo "pytest exit $?" | tee -a $S
tail -3 $OUT/pytest_gpu_$TAG.log | cut -c1-200 | tee -a $S
echo "== bench (default command)" | tee -a $S
timeout 900 python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
echo "bench exit $?" | tee -a $S
cut -c1-300 $OUT/bench_$TAG.json | tee -a $S
echo "== steady state (300 warm-up steps)" | tee -a $S
timeout 600 python bench.py --steps 50 --warmup 300 --no-cpu-baseline --no-kernel-timing --no-dcnv2 2>/dev/null | head -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('warm-up 300:', round(d['value']), d['ms_per_step'], d['step_us']['median'])" | tee -a $S
echo "== rocprofv3 kernel trace of the default command" | tee -a $S
rm -rf /tmp/prof_$TAG
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- \
    python $REPO/bench.py --no-cpu-baseline --no-kernel-timing --no-dcnv2 > $OUT/prof_bench_$TAG.json 2> $OUT/prof_$TAG.err)
STATS=$(ls -t $(find /tmp/prof_$TAG -name '*kernel_stats.csv') 2>/dev/null | head -1)
if [ -n "$STATS" ]; then cp $STATS $OUT/kernel_stats_deepfm_$TAG.csv; python scripts/kstats.py $STATS 20 10 | tee -a $S; fi
TR=$(find /tmp/prof_$TAG -name '*kernel_trace.csv' | head -1)
python scripts/step_timeline.py $TR 3 > $OUT/timeline_deepfm_$TAG.txt; tail -1 $OUT/timeline_deepfm_$TAG.txt | tee -a $S
grep catchup $OUT/timeline_deepfm_$TAG.txt | tee -a $S
