#!/bin/bash
# round 3, second full visit (after CIN on the matrix cores and DIN in the record): the whole GPU suite,
# the default bench line, rocprofv3 stats of the same command, one-step timelines + bench lines of every model.
TAG=${1:-r03final3}
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
S=$OUT/summary_$TAG.txt
echo "== smoke" | tee $S
timeout 600 python __graft_entry__.py --smoke > $OUT/smoke_$TAG.log 2>&1; echo "smoke exit $?" | tee -a $S
tail -1 $OUT/smoke_$TAG.log | tee -a $S
echo "== pytest -m gpu" | tee -a $S
FX_PARITY_REPORT=$OUT/parity_$TAG.jsonl timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -rs > $OUT/pytest_gpu_$TAG.log 2>&1
echo "pytest exit $?" | tee -a $S
tail -12 $OUT/pytest_gpu_$TAG.log | tee -a $S
echo "== bench (default command)" | tee -a $S
timeout 900 python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
echo "bench exit $?" | tee -a $S
cut -c1-330 $OUT/bench_$TAG.json | tee -a $S
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
for F in "--host-inputs" "--emb-dtype bf16"; do
  timeout 300 python bench.py $F --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-dcnv2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$F', round(d['value']), round(d['ms_per_step'],4))" | tee -a $S
done
FX_SHARD_WORLD1=1 timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-dcnv2 2>/dev/null | head -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('one RCCL rank, sharded', round(d['value']), round(d['ms_per_step'],4))" | tee -a $S
