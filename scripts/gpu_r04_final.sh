#!/bin/bash
# round 4, final visit: smoke, the whole GPU suite, the default bench line + the driver's settings, rocprofv3 kernel
# stats of the same command, one-step timelines + bench lines of every model, the one-rank sharded step, PMC traffic.
TAG=${1:-r04final}
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
S=$OUT/summary_$TAG.txt
echo "== smoke" | tee $S
timeout 600 python __graft_entry__.py --smoke > $OUT/smoke_$TAG.log 2>&1; echo "smoke exit $?" | tee -a $S
tail -1 $OUT/smoke_$TAG.log | cut -c1-300 | tee -a $S
echo "== pytest -m gpu" | tee -a $S
FX_PARITY_REPORT=$OUT/parity_$TAG.jsonl timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -rs > $OUT/pytest_gpu_$TAG.log 2>&1
echo "pytest exit $?" | tee -a $S
tail -6 $OUT/pytest_gpu_$TAG.log | cut -c1-200 | tee -a $S
echo "== bench (default command)" | tee -a $S
timeout 900 python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
echo "bench exit $?" | tee -a $S
cut -c1-400 $OUT/bench_$TAG.json | tee -a $S
echo "== bench (the driver's settings: --steps 20 --warmup 5)" | tee -a $S
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_$TAG.json 2>/dev/null
python - <<PY | tee -a $S
import json
d = json.load(open("$OUT/bench_driver_$TAG.json"))
print("driver settings:", round(d["value"]), d["ms_per_step"], "events", d.get("ms_per_step_events"), "step_us", d.get("step_us"),
      "kernel_sum_us", d.get("kernel_sum_us"), "wall-kernel", d.get("wall_minus_kernel_sum_us"))
x = d.get("dcnv2") or {}
print("dcnv2 block:", {k: x.get(k) for k in ("value", "ms_per_step", "kernel_sum_us", "wall_minus_kernel_sum_us")})
PY
echo "== steady state of the exact-mode catch-up (300 / 1000 warm-up steps)" | tee -a $S
timeout 600 python bench.py --steps 50 --warmup 1000 --no-cpu-baseline --no-kernel-timing --no-dcnv2 2>/dev/null | head -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('warm-up 1000:', round(d['value']), d['ms_per_step'], d['step_us']['median'])" | tee -a $S
timeout 600 python bench.py --steps 50 --warmup 300 --no-cpu-baseline --no-kernel-timing --no-dcnv2 2>/dev/null | head -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('warm-up 300:', round(d['value']), d['ms_per_step'], d['step_us'])" | tee -a $S
echo "== rocprofv3 kernel trace of the default command" | tee -a $S
rm -rf /tmp/prof_$TAG
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- \
    python $REPO/bench.py --no-cpu-baseline --no-kernel-timing --no-dcnv2 > $OUT/prof_bench_$TAG.json 2> $OUT/prof_$TAG.err)
STATS=$(ls -t $(find /tmp/prof_$TAG -name '*kernel_stats.csv') 2>/dev/null | head -1)
if [ -n "$STATS" ]; then cp $STATS $OUT/kernel_stats_deepfm_$TAG.csv; python scripts/kstats.py $STATS 20 10 | tee -a $S; fi
TR=$(find /tmp/prof_$TAG -name '*kernel_trace.csv' | head -1)
python scripts/step_timeline.py $TR 3 > $OUT/timeline_deepfm_$TAG.txt; tail -1 $OUT/timeline_deepfm_$TAG.txt | tee -a $S
for M in DCNv2 DIN DLRM xDeepFM; do
  rm -rf /tmp/prof_${TAG}_$M
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_$M -- \
      python $REPO/bench.py --model $M --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > /dev/null 2> $OUT/prof_${TAG}_$M.err)
  TR=$(find /tmp/prof_${TAG}_$M -name '*kernel_trace.csv' | head -1)
  python scripts/step_timeline.py $TR 3 > $OUT/timeline_${M}_$TAG.txt; echo "$M $(tail -1 $OUT/timeline_${M}_$TAG.txt)" | tee -a $S
  ST=$(ls -t $(find /tmp/prof_${TAG}_$M -name '*kernel_stats.csv') 2>/dev/null | head -1)
  if [ -n "$ST" ]; then cp $ST $OUT/kernel_stats_${M}_$TAG.csv; fi
  timeout 300 python bench.py --model $M --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench_${M}_$TAG.json 2>/dev/null
  python -c "import json; d=json.load(open('$OUT/bench_${M}_$TAG.json')); print('$M', round(d['value']), round(d['ms_per_step'],4), {k: round(v['frac'],3) for k,v in d.items() if k.startswith('roofline') and isinstance(v, dict) and 'frac' in v})" | tee -a $S
done
echo "== other modes (DeepFM)" | tee -a $S
for F in "--host-inputs" "--emb-dtype bf16" "--batch 32768"; do
  timeout 300 python bench.py $F --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-dcnv2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$F', round(d['value']), round(d['ms_per_step'],4))" | tee -a $S
done
echo "== one RCCL rank, row-sharded (recorded collectives; FX_GRAPH_COLLECTIVES=0: segments)" | tee -a $S
FX_SHARD_WORLD1=1 timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-dcnv2 2>/dev/null | head -1 > $OUT/bench_shard1_$TAG.json
python -c "import json; d=json.load(open('$OUT/bench_shard1_$TAG.json')); print('one RCCL rank, sharded', round(d['value']), round(d['ms_per_step'],4), d['config'].get('parallelism'))" | tee -a $S
FX_SHARD_WORLD1=1 FX_GRAPH_COLLECTIVES=0 timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-dcnv2 2>/dev/null | head -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('one RCCL rank, segments', round(d['value']), round(d['ms_per_step'],4), d['config'].get('parallelism'))" | tee -a $S
rm -rf /tmp/prof_${TAG}_shard
(cd /tmp && FX_SHARD_WORLD1=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_shard -- \
    python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-dcnv2 > /dev/null 2> $OUT/prof_${TAG}_shard.err)
TR=$(find /tmp/prof_${TAG}_shard -name '*kernel_trace.csv' | head -1)
python scripts/step_timeline.py $TR 3 > $OUT/timeline_deepfm_shard1_$TAG.txt; echo "sharded $(tail -1 $OUT/timeline_deepfm_shard1_$TAG.txt)" | tee -a $S
if [ -z "$SKIP_PMC" ]; then
echo "== PMC traffic" | tee -a $S
bash scripts/pmc_traffic.sh $TAG 2>&1 | tail -25 | cut -c1-200 | tee -a $S
fi
