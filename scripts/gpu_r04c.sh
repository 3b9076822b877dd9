#!/bin/bash
# round 4, visit c: the round-4 gather kernel (bit identity + A/B), first-step gradient parity at the c3 shapes,
# the one-rank sharded bench lines (segments / recorded collectives).
TAG=${1:-r04c}
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
S=$OUT/summary_$TAG.txt
echo "== pytest fused + shard kernels" | tee $S
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_shard_kernels.py -q --timeout 600 -p no:cacheprovider -x > $OUT/pytest_$TAG.log 2>&1
echo "pytest exit $?" | tee -a $S
tail -8 $OUT/pytest_$TAG.log | tee -a $S
echo "== gradient parity" | tee -a $S
rm -f $OUT/grad_parity_$TAG.jsonl
FX_GRAD_PARITY_REPORT=$OUT/grad_parity_$TAG.jsonl timeout 1500 python -m pytest tests/test_gpu_grad_parity.py -q --timeout 1200 -p no:cacheprovider -s > $OUT/pytest_grad_$TAG.log 2>&1
echo "pytest exit $?" | tee -a $S
tail -25 $OUT/pytest_grad_$TAG.log | cut -c1-600 | tee -a $S
python scripts/grad_parity_table.py $OUT/grad_parity_$TAG.jsonl > $OUT/grad_parity_table_$TAG.txt 2>&1; tail -40 $OUT/grad_parity_table_$TAG.txt | tee -a $S
echo "== gather A/B (FX_EMB_FWD2)" | tee -a $S
for R in 1 2; do for V in 1 0; do
  FX_EMB_FWD2=$V timeout 400 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-dcnv2 2>/dev/null | head -1 > $OUT/ab_tmp.json
  python -c "import json; d=json.load(open('$OUT/ab_tmp.json')); print('FX_EMB_FWD2=$V', round(d['value']), round(d['ms_per_step'],4), 'gather', round(d['roofline_gather']['avg_launch_us'],2), round(d['roofline_gather']['frac'],3), 'b32768', round(d['roofline_gather_b32768']['avg_launch_us'],2), round(d['roofline_gather_b32768']['frac'],3), 'sparse', round(d['roofline_sparse']['us_per_step'],1))" | tee -a $S
done; done
for M in DCNv2 DIN; do for V in 1 0; do
  FX_EMB_FWD2=$V timeout 400 python bench.py --model $M --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | head -1 > $OUT/ab_tmp.json
  python -c "import json; d=json.load(open('$OUT/ab_tmp.json')); print('$M FX_EMB_FWD2=$V', round(d['value']), round(d['ms_per_step'],4), d['step_us']['median'])" | tee -a $S
done; done
echo "== one RCCL rank, sharded: segments / recorded collectives (default)" | tee -a $S
for G in 0 1; do
  T0=$(date +%s.%N)
  FX_GRAPH_COLLECTIVES=$G FX_SHARD_WORLD1=1 timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-dcnv2 2>$OUT/shard_${G}_$TAG.err | head -1 > $OUT/bench_shard_g${G}_$TAG.json
  echo "exit ${PIPESTATUS[0]} wall $(python -c "import time; print(round(time.time()-$T0,1))") s" | tee -a $S
  python -c "import json; d=json.load(open('$OUT/bench_shard_g${G}_$TAG.json')); print('FX_GRAPH_COLLECTIVES=$G', round(d['value']), round(d['ms_per_step'],4), round(d['ms_per_step_events'],4), d['step_us']['median'], d['config']['parallelism'][-90:])" | tee -a $S
done
timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-dcnv2 2>/dev/null | head -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('unsharded, same box', round(d['value']), round(d['ms_per_step'],4), d['step_us']['median'])" | tee -a $S
