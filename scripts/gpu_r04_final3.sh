#!/bin/bash
# round 4, last visit (the catch-up replay's window test): whole GPU suite, default bench line, driver settings,
# steady state, DeepFM kernel stats + timeline, bench lines of the other models, one-rank sharded line.
TAG=${1:-r04final3}
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
S=$OUT/summary_$TAG.txt
echo "== smoke" | tee $S
timeout 600 python __graft_entry__.py --smoke > $OUT/smoke_$TAG.log 2>&1; echo "smoke exit $?" | tee -a $S
tail -1 $OUT/smoke_$TAG.log | cut -c1-300 | tee -a $S
echo "== pytest -m gpu" | tee -a $S
FX_PARITY_REPORT=$OUT/parity_$TAG.jsonl timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -rs > $OUT/pytest_gpu_$TAG.log 2>&1
echo "pytest exit $?" | tee -a $S
tail -3 $OUT/pytest_gpu_$TAG.log | cut -c1-200 | tee -a $S
echo "== bench (default command)" | tee -a $S
timeout 900 python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
echo "bench exit $?" | tee -a $S
cut -c1-400 $OUT/bench_$TAG.json | tee -a $S
echo "== bench (the driver's settings: --steps 20 --warmup 5)" | tee -a $S
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_$TAG.json 2>/dev/null
python - <<PY | tee -a $S
import json
d = json.load(open("$OUT/bench_driver_$TAG.json"))
print("driver settings:", round(d["value"]), d["ms_per_step"], "events", d.get("ms_per_step_events"), "step_us median", d["step_us"]["median"],
      "kernel_sum_us", d.get("kernel_sum_us"), "wall-kernel", d.get("wall_minus_kernel_sum_us"))
x = d.get("dcnv2") or {}
print("dcnv2 block:", {k: x.get(k) for k in ("value", "ms_per_step", "kernel_sum_us", "wall_minus_kernel_sum_us")})
PY
echo "== steady state of the exact-mode catch-up (300 warm-up steps)" | tee -a $S
timeout 600 python bench.py --steps 50 --warmup 300 --no-cpu-baseline --no-kernel-timing --no-dcnv2 2>/dev/null | head -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('warm-up 300:', round(d['value']), d['ms_per_step'], d['step_us']['median'])" | tee -a $S
echo "== rocprofv3 kernel trace of the default command" | tee -a $S
rm -rf /tmp/prof_$TAG
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- \
    python $REPO/bench.py --no-cpu-baseline --no-kernel-timing --no-dcnv2 > $OUT/prof_bench_$TAG.json 2> $OUT/prof_$TAG.err)
STATS=$(ls -t $(find /tmp/prof_$TAG -name '*kernel_stats.csv') 2>/dev/null | head -1)
if [ -n "$STATS" ]; then cp $STATS $OUT/kernel_stats_deepfm_$TAG.csv; python scripts/kstats.py $STATS 12 10 | tee -a $S; fi
TR=$(find /tmp/prof_$TAG -name '*kernel_trace.csv' | head -1)
python scripts/step_timeline.py $TR 3 > $OUT/timeline_deepfm_$TAG.txt; tail -1 $OUT/timeline_deepfm_$TAG.txt | tee -a $S
grep catchup $OUT/timeline_deepfm_$TAG.txt | tee -a $S
for M in DCNv2 DIN DLRM xDeepFM; do
  timeout 300 python bench.py --model $M --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench_${M}_$TAG.json 2>/dev/null
  python -c "import json; d=json.load(open('$OUT/bench_${M}_$TAG.json')); print('$M', round(d['value']), round(d['ms_per_step'],4), {k: round(v['frac'],3) for k,v in d.items() if k.startswith('roofline') and isinstance(v, dict) and 'frac' in v})" | tee -a $S
done
FX_SHARD_WORLD1=1 timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-dcnv2 2>/dev/null | head -1 > $OUT/bench_shard1_$TAG.json
python -c "import json; d=json.load(open('$OUT/bench_shard1_$TAG.json')); print('one RCCL rank, sharded', round(d['value']), round(d['ms_per_step'],4))" | tee -a $S
