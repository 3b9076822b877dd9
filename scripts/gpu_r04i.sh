#!/bin/bash
# round 4, visit i: the driver's default command on the sharded path (kernel-timing pass included), the capture-failure
# fallback test, the dist tests, model benches
TAG=${1:-r04i}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
S=$OUT/summary_$TAG.txt
echo "== pytest dist" | tee $S
timeout 1200 python -m pytest tests/test_gpu_dist.py -q --timeout 900 -p no:cacheprovider -x > $OUT/pytest_$TAG.log 2>&1
echo "pytest exit $?" | tee -a $S; tail -4 $OUT/pytest_$TAG.log | cut -c1-300 | tee -a $S
echo "== sharded one-rank run with the DEFAULT flags of the driver (kernel timing pass on)" | tee -a $S
T0=$(date +%s)
FX_SHARD_WORLD1=1 timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_shard_default_$TAG.json 2> $OUT/bench_shard_default_$TAG.err
echo "exit $? wall $(( $(date +%s) - T0 )) s" | tee -a $S
tail -3 $OUT/bench_shard_default_$TAG.err | cut -c1-300 | tee -a $S
python -c "
import json; d=json.load(open('$OUT/bench_shard_default_$TAG.json'))
print(round(d['value']), round(d['ms_per_step'],4), d['config']['parallelism'][-80:])
print({k: (round(v['frac'],3) if isinstance(v, dict) and 'frac' in v else None) for k, v in d.items() if k.startswith('roofline')})
print('kernel_sum', d.get('kernel_sum_us'), d.get('kernel_sum_launches'))" 2>&1 | tee -a $S
echo "== gloo two ranks on one GPU, default flags" | tee -a $S
T0=$(date +%s)
FX_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 10 --warmup 5 --vocab-scale 0.05 --no-cpu-baseline > $OUT/bench_gloo2_$TAG.json 2> $OUT/bench_gloo2_$TAG.err
echo "exit $? wall $(( $(date +%s) - T0 )) s" | tee -a $S
tail -2 $OUT/bench_gloo2_$TAG.err | cut -c1-300 | tee -a $S
python -c "
import json; d=json.load(open('$OUT/bench_gloo2_$TAG.json'))
print(d['n_gpus'], round(d['value']), round(d['ms_per_step'],4), d['config']['parallelism'][-80:])" 2>&1 | tee -a $S
