#!/bin/bash
# round 4, visit a: where the driver's wall clock goes (event pair, per-step spread, kernel sum at the driver's
# 20/5 settings), the launch floor of a graph node, and the one-rank sharded step's timeline before this round's work.
TAG=${1:-r04a}
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
S=$OUT/summary_$TAG.txt
echo "== launch floor" | tee $S
timeout 120 scripts/ubench/launch_floor 2>&1 | tee $OUT/launch_floor_$TAG.txt | tee -a $S
echo "== bench at the driver's settings (20 steps, 5 warm-up)" | tee -a $S
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_$TAG.json 2> $OUT/bench_driver_$TAG.err
echo "exit $?" | tee -a $S
python - <<PY | tee -a $S
import json
d = json.load(open("$OUT/bench_driver_$TAG.json"))
for name, o in (("DeepFM", d), ("DCNv2", d.get("dcnv2", {}))):
    print(name, {k: o.get(k) for k in ("value", "ms_per_step", "ms_per_step_events", "warmup_steps_run",
                                        "kernel_sum_us", "kernel_sum_launches", "wall_minus_kernel_sum_us")})
    print("   step_us", o.get("step_us"))
    print("   rooflines", {k: round(v["frac"], 3) for k, v in o.items() if k.startswith("roofline") and isinstance(v, dict)})
PY
echo "== default bench (50/10)" | tee -a $S
timeout 600 python bench.py --no-cpu-baseline --no-dcnv2 --no-kernel-timing > $OUT/bench_default_$TAG.json 2>/dev/null
python -c "import json; d=json.load(open('$OUT/bench_default_$TAG.json')); print(round(d['value']), round(d['ms_per_step'],4), round(d['ms_per_step_events'],4), d['step_us'])" | tee -a $S
echo "== one RCCL rank, sharded: segments / recorded collectives" | tee -a $S
for G in 0 1; do
  FX_GRAPH_COLLECTIVES=$G FX_SHARD_WORLD1=1 timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-dcnv2 2>$OUT/shard_$G_$TAG.err | head -1 > $OUT/bench_shard_g${G}_$TAG.json
  python -c "import json; d=json.load(open('$OUT/bench_shard_g${G}_$TAG.json')); print('FX_GRAPH_COLLECTIVES=$G', round(d['value']), round(d['ms_per_step'],4), round(d['ms_per_step_events'],4), d['step_us'])" | tee -a $S
done
echo "== rocprofv3 timeline of the sharded step (segments)" | tee -a $S
rm -rf /tmp/prof_sh
(cd /tmp && FX_SHARD_WORLD1=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sh -- \
    python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-dcnv2 --no-step-events > /dev/null 2> $OUT/prof_sh_$TAG.err)
TR=$(find /tmp/prof_sh -name '*kernel_trace.csv' | head -1)
python scripts/step_timeline.py $TR 3 > $OUT/timeline_deepfm_shard_world1_$TAG.txt; tail -1 $OUT/timeline_deepfm_shard_world1_$TAG.txt | tee -a $S
echo "== rocprofv3 timeline of the unsharded step (same box)" | tee -a $S
rm -rf /tmp/prof_un
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_un -- \
    python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-dcnv2 --no-step-events > /dev/null 2> $OUT/prof_un_$TAG.err)
TR=$(find /tmp/prof_un -name '*kernel_trace.csv' | head -1)
python scripts/step_timeline.py $TR 3 > $OUT/timeline_deepfm_$TAG.txt; tail -1 $OUT/timeline_deepfm_$TAG.txt | tee -a $S
