#!/bin/bash
# round 4, visit p: labels asked for after get_inputs() staged the batch — whole GPU suite both ways, step A/B, timelines
TAG=${1:-r04p}
OUT=$PWD/gpurun_out; mkdir -p $OUT
S=$OUT/head_train_$TAG.txt; : > $S
echo "== trajectory tests, FX_HEAD_FUSED=0" | tee -a $S
FX_HEAD_FUSED=0 timeout 900 python -m pytest tests/test_gpu_models.py -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | grep -E "^FAILED|passed|failed" | tee -a $S
echo "== pytest -m gpu (whole suite)" | tee -a $S
FX_PARITY_REPORT=$OUT/parity_$TAG.jsonl timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -rs > $OUT/pytest_gpu_$TAG.log 2>&1
echo "pytest exit $?" | tee -a $S
grep -E "^FAILED|passed|failed" $OUT/pytest_gpu_$TAG.log | tee -a $S
echo "== step A/B (median step_us, value)" | tee -a $S
for R in 1 2; do for E in 0 1; do for M in DeepFM DCNv2 DIN DLRM xDeepFM; do
  FX_HEAD_FUSED=$E timeout 400 python bench.py --model $M --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-dcnv2 2>/dev/null | head -1 > $OUT/ab_tmp.json
  python -c "import json; d=json.load(open('$OUT/ab_tmp.json')); print('$M', 'head_fused=$E', round(d['value']), round(d['ms_per_step'],4), d['step_us']['median'])" 2>&1 | tail -1 | tee -a $S
done; done; done
echo "== step timelines" | tee -a $S
REPO=$PWD; export TMPDIR=/tmp
for M in DeepFM DCNv2; do
rm -rf /tmp/prof_${TAG}_$M
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_$M -- \
    python $REPO/bench.py --model $M --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-dcnv2 > /dev/null 2> $OUT/prof_${TAG}_$M.err)
TR=$(find /tmp/prof_${TAG}_$M -name '*kernel_trace.csv' | head -1)
python scripts/step_timeline.py $TR 3 > $OUT/timeline_${M}_$TAG.txt; cat $OUT/timeline_${M}_$TAG.txt | tee -a $S
done
