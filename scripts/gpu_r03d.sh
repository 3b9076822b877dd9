#!/bin/bash
# round 3, visit d: register-based fused row sums, multi grid as the default pair path; lab + tests + bench
TAG=${1:-r03d}
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
S=$OUT/summary_$TAG.txt
LAB=$REPO/scripts/ubench/gemm_lab
L=$OUT/gemm_lab_$TAG.txt; : > $L
timeout 200 $LAB all >> $L 2>&1
timeout 200 $LAB mix >> $L 2>&1
timeout 200 $LAB pairs --check >> $L 2>&1
FX_GEMM_MULTI=0 FX_LAB_TAG=" multi=0" timeout 120 $LAB pairs --check >> $L 2>&1
FX_GEMM_TILE=128x128 timeout 120 $LAB tower --trace >> $L 2>&1
FX_MULTI_CFG="0,4;0,4" FX_LAB_TAG=" cfg=0,4;0,4" timeout 200 $LAB mix --trace >> $L 2>&1
for TILE in 64x64 128x64 128x128; do
  FX_GEMM_TILE=$TILE timeout 300 $LAB odd --check >> $L 2>&1
done
timeout 300 $LAB tower --check >> $L 2>&1
timeout 300 $LAB cross --check >> $L 2>&1
grep -c MISMATCH $L | sed 's/^/MISMATCH lines: /' | tee $S
grep "^\[" $L | head -60 | tee -a $S
echo "== pytest subset" | tee -a $S
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_baseline_shapes.py -m gpu -q -x --timeout 600 -p no:cacheprovider > $OUT/pytest_subset_$TAG.log 2>&1
echo "pytest exit $?" | tee -a $S
tail -5 $OUT/pytest_subset_$TAG.log | tee -a $S
echo "== bench (default command)" | tee -a $S
timeout 900 python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
echo "bench exit $?" | tee -a $S
python - <<PY | tee -a $S
import json
d = json.loads(open("$OUT/bench_$TAG.json").readline())
print("value", round(d["value"]), "ms", round(d["ms_per_step"], 4), "roofline", round(d["roofline"]["frac"], 3))
print("dcnv2", round(d["dcnv2"]["value"]), round(d["dcnv2"]["ms_per_step"], 4), round(d["dcnv2"]["roofline"]["frac"], 3))
for k, v in d["roofline"]["by_shape_MxNxK"].items(): print("  ", k, v)
for k, v in d["dcnv2"]["roofline"]["by_shape_MxNxK"].items(): print("  dcnv2", k, v)
PY
for M in DIN DLRM xDeepFM; do
  timeout 300 python bench.py --model $M --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing > $OUT/bench_${M}_$TAG.json 2>/dev/null
  python -c "import json; d=json.load(open('$OUT/bench_${M}_$TAG.json')); print('$M', round(d['value']), round(d['ms_per_step'],4))" | tee -a $S
done
