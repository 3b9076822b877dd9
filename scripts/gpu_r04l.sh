#!/bin/bash
# round 4, visit l: the K-slab reduce on 16-byte vectors (FX_SPLITK_V4, default 1) — lab A/B, tests, step A/B
TAG=${1:-r04l}
OUT=$PWD/gpurun_out; mkdir -p $OUT
S=$OUT/splitk_v4_$TAG.txt; : > $S
L=scripts/ubench/gemm_lab
for E in 0 1; do
  for SU in pairs; do
    echo "--- FX_SPLITK_V4=$E suite $SU" | tee -a $S
    FX_SPLITK_V4=$E FX_LAB_TAG=" [splitk_v4=$E]" timeout 200 $L $SU 2>&1 | grep -v "^$" | cut -c1-200 | tee -a $S
  done
done
echo "--- suite tower / cross / odd --check" | tee -a $S
for SU in tower cross odd; do timeout 200 $L $SU --check 2>&1 | grep -v "^$" | cut -c1-200 | tee -a $S; done
echo "== GEMM / tower / cross tests" | tee -a $S
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider -k "gemm or mlp or cross or tower or linear or multi or pair or slab" 2>&1 | tail -4 | tee -a $S
echo "== step A/B (median step_us, value)" | tee -a $S
for R in 1 2; do for E in 0 1; do for M in DeepFM DCNv2 DIN DLRM xDeepFM; do
  FX_SPLITK_V4=$E timeout 400 python bench.py --model $M --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-dcnv2 2>/dev/null | head -1 > $OUT/ab_tmp.json
  python -c "import json; d=json.load(open('$OUT/ab_tmp.json')); print('$M', 'splitk_v4=$E', round(d['value']), round(d['ms_per_step'],4), d['step_us']['median'])" 2>&1 | tail -1 | tee -a $S
done; done; done
echo "== rocprof kernel stats DeepFM" | tee -a $S
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o deepfm -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-dcnv2 > $OUT/prof_$TAG.log 2>&1
cd $GRAFT_REPO_ROOT; python - <<PY | tee -a $S
import csv, glob
f = glob.glob("$OUT/prof_$TAG/**/*kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    for r in rows[:24]:
        print("%-60s calls %6s avg %9.1f ns  %5.1f%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]), float(r["Percentage"])))
PY
