#!/bin/bash
# round 3, visit e: DCNv2 parallel structure — cross layer + deep layer of one depth as one grid
TAG=${1:-r03e}
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
S=$OUT/summary_$TAG.txt
LAB=$REPO/scripts/ubench/gemm_lab
L=$OUT/gemm_lab_$TAG.txt; : > $L
for R in 1 2; do
  FX_LAB_TAG=" multi=1 run$R" timeout 200 $LAB dcn >> $L 2>&1
  FX_GEMM_MULTI=0 FX_LAB_TAG=" multi=0 run$R" timeout 200 $LAB dcn >> $L 2>&1
done
grep "^\[" $L | sort -k6,6 -s | tee $S
echo "== pytest (DCNv2 paths)" | tee -a $S
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_c1_tiny_npz.py tests/test_gpu_kernels.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "dcnv2 or DCNv2 or gemm or mask" > $OUT/pytest_subset_$TAG.log 2>&1
echo "pytest exit $?" | tee -a $S
tail -4 $OUT/pytest_subset_$TAG.log | tee -a $S
timeout 600 python -m pytest tests/test_gpu_baseline_shapes.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "c3" > $OUT/pytest_c3_$TAG.log 2>&1
echo "pytest c3 exit $?" | tee -a $S
tail -3 $OUT/pytest_c3_$TAG.log | tee -a $S
echo "== bench DCNv2" | tee -a $S
for R in 1 2; do
  timeout 300 python bench.py --model DCNv2 --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench_DCNv2_$TAG.json 2>$OUT/bench_DCNv2_$TAG.err
  python - <<PY | tee -a $S
import json
d = json.loads(open("$OUT/bench_DCNv2_$TAG.json").readline())
print("DCNv2 fused", round(d["value"]), round(d["ms_per_step"], 4), round(d["roofline"]["frac"], 3))
for k, v in d["roofline"]["by_shape_MxNxK"].items(): print("  ", k, v)
PY
  FX_GEMM_MULTI=0 timeout 300 python bench.py --model DCNv2 --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing > $OUT/bench_DCNv2_m0_$TAG.json 2>/dev/null
  python -c "import json; d=json.load(open('$OUT/bench_DCNv2_m0_$TAG.json')); print('DCNv2 FX_GEMM_MULTI=0', round(d['value']), round(d['ms_per_step'],4))" | tee -a $S
done
timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-dcnv2 > $OUT/bench_DeepFM_$TAG.json 2>/dev/null
python -c "import json; d=json.load(open('$OUT/bench_DeepFM_$TAG.json')); print('DeepFM', round(d['value']), round(d['ms_per_step'],4))" | tee -a $S
FX_GEMM_MULTI=0 timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-dcnv2 > $OUT/bench_DeepFM_m0_$TAG.json 2>/dev/null
python -c "import json; d=json.load(open('$OUT/bench_DeepFM_m0_$TAG.json')); print('DeepFM FX_GEMM_MULTI=0', round(d['value']), round(d['ms_per_step'],4))" | tee -a $S
