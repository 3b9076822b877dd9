#!/bin/bash
# round 4, visit b: the one-launch kernels of the sharded step, the recorded-collectives default and its
# teardown, the one-rank sharded timeline after the rework, the default bench with the warm-up up to the clock.
TAG=${1:-r04b}
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
S=$OUT/summary_$TAG.txt
echo "== pytest shard kernels + dist" | tee $S
timeout 900 python -m pytest tests/test_gpu_shard_kernels.py tests/test_gpu_dist.py tests/test_gpu_fused.py -q --timeout 600 -p no:cacheprovider -x > $OUT/pytest_$TAG.log 2>&1
echo "pytest exit $?" | tee -a $S
tail -15 $OUT/pytest_$TAG.log | tee -a $S
echo "== bench at the driver's settings" | tee -a $S
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_$TAG.json 2> $OUT/bench_driver_$TAG.err
echo "exit $?" | tee -a $S
python - <<PY | tee -a $S
import json
d = json.load(open("$OUT/bench_driver_$TAG.json"))
for name, o in (("DeepFM", d), ("DCNv2", d.get("dcnv2", {}))):
    print(name, {k: o.get(k) for k in ("value", "ms_per_step", "ms_per_step_events", "warmup_steps_run",
                                        "kernel_sum_us", "wall_minus_kernel_sum_us")})
    print("   step_us", o.get("step_us"))
PY
echo "== one RCCL rank, sharded: segments / recorded collectives (default)" | tee -a $S
for G in 0 1; do
  /usr/bin/time -f "wall %e s" env FX_GRAPH_COLLECTIVES=$G FX_SHARD_WORLD1=1 timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-dcnv2 2>$OUT/shard_${G}_$TAG.err | head -1 > $OUT/bench_shard_g${G}_$TAG.json
  echo "exit ${PIPESTATUS[0]}" | tee -a $S; tail -2 $OUT/shard_${G}_$TAG.err | tee -a $S
  python -c "import json; d=json.load(open('$OUT/bench_shard_g${G}_$TAG.json')); print('FX_GRAPH_COLLECTIVES=$G', round(d['value']), round(d['ms_per_step'],4), round(d['ms_per_step_events'],4), d['step_us'], d['config']['parallelism'][-90:])" | tee -a $S
done
echo "== rocprofv3 timeline of the sharded step (recorded collectives)" | tee -a $S
rm -rf /tmp/prof_sh
(cd /tmp && FX_SHARD_WORLD1=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sh -- \
    python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-dcnv2 --no-step-events > /dev/null 2> $OUT/prof_sh_$TAG.err)
TR=$(find /tmp/prof_sh -name '*kernel_trace.csv' | head -1)
python scripts/step_timeline.py $TR 3 > $OUT/timeline_deepfm_shard_world1_$TAG.txt; tail -1 $OUT/timeline_deepfm_shard_world1_$TAG.txt | tee -a $S
