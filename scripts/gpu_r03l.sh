#!/bin/bash
# two-stream overlap (dW of the first layer beside the embedding backward; dense update beside the row update)
TAG=${1:-r03p}
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
S=$OUT/summary_$TAG.txt
echo "== pytest models+fused" | tee $S
timeout 1500 python -m pytest tests/test_gpu_models.py tests/test_gpu_fused.py -m gpu -q -x --timeout 900 -p no:cacheprovider > $OUT/pytest_$TAG.log 2>&1
echo "pytest exit $?" | tee -a $S
tail -5 $OUT/pytest_$TAG.log | tee -a $S
for M in DeepFM DIN DLRM; do
  echo "== A/B overlap ($M)" | tee -a $S
  bash scripts/gpu_ab.sh ovl_${M}_$TAG $M "FX_OVERLAP=1" "FX_OVERLAP=0" 2 | tee -a $S
done
M=DeepFM
rm -rf /tmp/prof_${TAG}_$M
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_$M -- \
    python $REPO/bench.py --model $M --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-dcnv2 > /dev/null 2> $OUT/prof_${TAG}_$M.err)
TR=$(find /tmp/prof_${TAG}_$M -name '*kernel_trace.csv' | head -1)
python scripts/step_timeline.py $TR 3 > $OUT/timeline_${M}_$TAG.txt
tail -16 $OUT/timeline_${M}_$TAG.txt | tee -a $S
