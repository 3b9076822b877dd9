#!/bin/bash
TAG=${1:-r03i}
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
S=$OUT/summary_$TAG.txt
echo "== pytest" | tee $S
timeout 1200 python -m pytest tests/test_gpu_fused.py tests/test_gpu_models.py tests/test_gpu_dist.py -m gpu -q -x --timeout 900 -p no:cacheprovider > $OUT/pytest_subset_$TAG.log 2>&1
echo "pytest exit $?" | tee -a $S
tail -6 $OUT/pytest_subset_$TAG.log | tee -a $S
echo "== A/B split catch-up (DeepFM)" | tee -a $S
bash scripts/gpu_ab.sh splitcatch_$TAG DeepFM "FX_SPLIT_CATCHUP=1" "FX_SPLIT_CATCHUP=0" 3 | tee -a $S
cp $OUT/ab_splitcatch_$TAG.txt $OUT/sparse_ab_$TAG.txt
for M in DeepFM DCNv2 DIN DLRM; do
  rm -rf /tmp/prof_${TAG}_$M
  EXTRA=""; if [ $M = DeepFM ]; then EXTRA="--no-dcnv2"; fi
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_${TAG}_$M -- \
      python $REPO/bench.py --model $M --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing $EXTRA > /dev/null 2> $OUT/prof_${TAG}_$M.err)
  TR=$(find /tmp/prof_${TAG}_$M -name '*kernel_trace.csv' | head -1)
  python scripts/step_timeline.py $TR 3 > $OUT/timeline_${M}_$TAG.txt; echo "$M $(tail -1 $OUT/timeline_${M}_$TAG.txt)" | tee -a $S
done
cat $OUT/timeline_DCNv2_$TAG.txt | tee -a $S
