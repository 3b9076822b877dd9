#!/bin/bash
# round 4, visit m: 16-slab reduce with all loads in flight (the DIN tower's weight gradients) — tests, step A/B, DIN timeline
TAG=${1:-r04m}
OUT=$PWD/gpurun_out; mkdir -p $OUT
S=$OUT/splitk_v4b_$TAG.txt; : > $S
echo "== GEMM tests" | tee -a $S
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider -k "gemm or mlp or cross or tower or linear or multi or pair or slab or din" 2>&1 | tail -4 | tee -a $S
echo "== step A/B (median step_us, value)" | tee -a $S
for R in 1 2; do for E in 0 1; do for M in DIN DeepFM DLRM; do
  FX_SPLITK_V4=$E timeout 400 python bench.py --model $M --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-dcnv2 2>/dev/null | head -1 > $OUT/ab_tmp.json
  python -c "import json; d=json.load(open('$OUT/ab_tmp.json')); print('$M', 'splitk_v4=$E', round(d['value']), round(d['ms_per_step'],4), d['step_us']['median'])" 2>&1 | tail -1 | tee -a $S
done; done; done
echo "== DIN step timeline" | tee -a $S
REPO=$PWD; export TMPDIR=/tmp; rm -rf /tmp/prof_$TAG
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- \
    python $REPO/bench.py --model DIN --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > /dev/null 2> $OUT/prof_$TAG.err)
TR=$(find /tmp/prof_$TAG -name '*kernel_trace.csv' | head -1)
python scripts/step_timeline.py $TR 3 > $OUT/timeline_DIN_$TAG.txt; cat $OUT/timeline_DIN_$TAG.txt | tee -a $S
