#!/bin/bash
# round 2, visit F: dW+dX pair launch — GEMM tests, model parity, A/B bench (FX_GEMM_PAIR), timelines
TAG=${1:-r02f}
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
S=$OUT/summary_$TAG.txt
echo "== pytest gemm + models" | tee $S
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_fused.py -m gpu -q --timeout 600 -p no:cacheprovider > $OUT/pytest_$TAG.log 2>&1
echo "pytest exit $?" | tee -a $S
tail -12 $OUT/pytest_$TAG.log | tee -a $S
for V in 1 0; do
  for M in DeepFM DCNv2; do
    echo "== bench $M FX_GEMM_PAIR=$V" | tee -a $S
    FX_GEMM_PAIR=$V timeout 600 python bench.py --model $M --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-dcnv2 > $OUT/bench_${M}_pair${V}_$TAG.json 2> $OUT/bench_${M}_pair${V}_$TAG.err
    python -c "import json,sys; d=json.load(open('$OUT/bench_${M}_pair${V}_$TAG.json')); print(d['value'], d['ms_per_step'])" | tee -a $S
  done
done
for M in DeepFM DCNv2; do
  echo "== timeline $M" | tee -a $S
  rm -rf /tmp/prof_${TAG}_$M
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_$M -- \
      python $REPO/bench.py --model $M --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-dcnv2 > $OUT/prof_bench_${TAG}_$M.json 2> $OUT/prof_${TAG}_$M.err)
  STATS=$(ls -t $(find /tmp/prof_${TAG}_$M -name '*kernel_stats.csv') 2>/dev/null | head -1)
  if [ -n "$STATS" ]; then cp $STATS $OUT/kernel_stats_${M}_$TAG.csv; fi
  TR=$(find /tmp/prof_${TAG}_$M -name '*kernel_trace.csv' | head -1)
  python scripts/step_timeline.py $TR 3 > $OUT/timeline_${M}_$TAG.txt; grep -E "gemm_f32|splitk_reduce |kernels" $OUT/timeline_${M}_$TAG.txt | cut -c1-110 | tee -a $S
done
