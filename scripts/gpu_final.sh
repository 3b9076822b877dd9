#!/bin/bash
# A round's closing visit: the whole GPU suite, the driver's bench command, the other models' lines, the one-rank
# RCCL sharded step, rocprofv3 kernel stats + step timelines, PMC traffic.  usage: gpu_final.sh TAG
TAG=${1:-final}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
S=$OUT/${TAG}_final_visit_summary.txt; : > $S
echo "== pytest -m gpu (whole suite)" | tee -a $S
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -rs > $OUT/pytest_gpu_$TAG.log 2>&1
echo "pytest exit $?" | tee -a $S
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_gpu_$TAG.log | tail -20 | tee -a $S
grep -E "^E  " $OUT/pytest_gpu_$TAG.log | head -20 | cut -c1-300 | tee -a $S
echo "== python bench.py --gpus 1 --steps 20 --warmup 5 (the driver's command)" | tee -a $S
T0=$(date +%s); timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench_driver_settings_final.json 2> $OUT/bench_$TAG.err
echo "exit $? wall $(( $(date +%s) - T0 )) s" | tee -a $S
python - $OUT/${TAG}_bench_driver_settings_final.json <<'PY' | tee -a $S
import json, sys
d = json.load(open(sys.argv[1]))
def show(tag, x):
    r = x.get("roofline", {})
    print("%s: %.0f samples/s  %.4f ms/step steady state (young run %.4f)  gemm %.1f us/step, %.1f TFLOP/s = %.3f of %.1f" % (
        tag, x["value"], x["ms_per_step"], x.get("young_run", {}).get("ms_per_step", 0), r.get("gemm_us_per_step", 0), r.get("achieved", 0), r.get("frac", 0), r.get("peak", 0)))
    for k, v in sorted(r.get("by_shape_MxNxK", {}).items()):
        print("    %-64s x%.0f  %7.2f us  %6.1f TF  %.3f" % (k[:64], v["launches_per_step"], v["avg_launch_us"], v["tflops"], v["frac"]))
    sp = x.get("roofline_sparse")
    if sp: print("    sparse path: %.1f us/step, %.0f GB/s (%.3f of HBM)" % (sp["us_per_step"], sp["achieved"], sp["frac"]))
    for k in ("roofline_gather", "roofline_gather_b32768"):
        if k in x: print("    %s: %.1f us, %.0f GB/s (%.3f)" % (k, x[k]["avg_launch_us"], x[k]["achieved"], x[k]["frac"]))
    print("    kernel sum %s us in %s launches, wall - kernel sum %s us, warmup %s" % (x.get("kernel_sum_us"), x.get("kernel_sum_launches"), x.get("wall_minus_kernel_sum_us"), x.get("warmup_steps_run")))
show("DeepFM", d)
show("DCNv2", d["dcnv2"])
print("uniform ids:", round(d["value_uniform"]), "samples/s", round(d["uniform"]["ms_per_step"], 4), "ms")
pf = d["parity_full_vocab"]
print("parity_full_vocab: max_dlogit", pf["max_dlogit"], {k: {kk: v[kk] for kk in ("max_dlogit_before", "max_dlogit_after", "max_dloss", "max_dloss_yardstick")} for k, v in pf.items() if isinstance(v, dict)})
print("cpu_baseline:", round(d["cpu_baseline"]["value"]), "samples/s on", d["cpu_baseline"]["cores"], "threads")
PY
echo "== other models (50 steps, steady state)" | tee -a $S
for M in DIN DLRM xDeepFM; do
  timeout 600 python bench.py --model $M --steps 50 --warmup 10 --no-cpu-baseline > $OUT/${TAG}_bench_${M}_final.json 2>> $OUT/bench_$TAG.err
  python -c "import json; d=json.load(open('$OUT/${TAG}_bench_${M}_final.json')); print('$M', round(d['value']), 'samples/s', round(d['ms_per_step'],4), 'ms steady (young', d.get('young_run',{}).get('ms_per_step'), ')', {k: round(v['frac'],3) for k,v in d.items() if k.startswith('roofline') and isinstance(v, dict) and 'frac' in v})" | tee -a $S
done
echo "== row-sharded step on one RCCL rank" | tee -a $S
FX_SHARD_WORLD1=1 timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing > $OUT/${TAG}_bench_deepfm_rccl_world1_final.json 2>> $OUT/bench_$TAG.err
python -c "import json; d=json.load(open('$OUT/${TAG}_bench_deepfm_rccl_world1_final.json')); print('one RCCL rank:', round(d['value']), 'samples/s', round(d['ms_per_step'],4), 'ms;', d['config']['parallelism'][-90:])" | tee -a $S
echo "== rocprofv3 kernel traces" | tee -a $S
bash scripts/gpu_timelines.sh $TAG DeepFM DCNv2 DIN | tee -a $S
echo "== PMC traffic" | tee -a $S
bash scripts/pmc_traffic.sh $TAG | tail -40 | tee -a $S
