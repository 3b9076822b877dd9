#!/bin/bash
TAG=${1:-r03g}
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
S=$OUT/summary_$TAG.txt
LAB=$REPO/scripts/ubench/gemm_lab
L=$OUT/gemm_lab_$TAG.txt; : > $L
for R in 1 2; do
  FX_LAB_TAG=" rules run$R" timeout 200 $LAB dcn >> $L 2>&1
  FX_GEMM_MULTI=0 FX_LAB_TAG=" multi=0 run$R" timeout 200 $LAB dcn >> $L 2>&1
  FX_LAB_TAG=" rules run$R" timeout 200 $LAB pairs --check >> $L 2>&1
  FX_GEMM_MULTI=0 FX_LAB_TAG=" multi=0 run$R" timeout 200 $LAB pairs >> $L 2>&1
done
grep -c MISMATCH $L | sed 's/^/MISMATCH lines: /' | tee $S
grep "^\[" $L | grep "pair" | sort -k6,6 -s | tee -a $S
for R in 1 2; do
  timeout 300 python bench.py --model DCNv2 --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing > $OUT/bench_DCNv2_$TAG.json 2>$OUT/bench_DCNv2_$TAG.err
  python -c "import json; d=json.load(open('$OUT/bench_DCNv2_$TAG.json')); print('DCNv2', round(d['value']), round(d['ms_per_step'],4))" | tee -a $S
  FX_GEMM_MULTI=0 timeout 300 python bench.py --model DCNv2 --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing > $OUT/bench_DCNv2_m0_$TAG.json 2>/dev/null
  python -c "import json; d=json.load(open('$OUT/bench_DCNv2_m0_$TAG.json')); print('DCNv2 FX_GEMM_MULTI=0', round(d['value']), round(d['ms_per_step'],4))" | tee -a $S
  timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-dcnv2 > $OUT/bench_DeepFM_$TAG.json 2>/dev/null
  python -c "import json; d=json.load(open('$OUT/bench_DeepFM_$TAG.json')); print('DeepFM', round(d['value']), round(d['ms_per_step'],4))" | tee -a $S
  FX_GEMM_MULTI=0 timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-dcnv2 > $OUT/bench_DeepFM_m0_$TAG.json 2>/dev/null
  python -c "import json; d=json.load(open('$OUT/bench_DeepFM_m0_$TAG.json')); print('DeepFM FX_GEMM_MULTI=0', round(d['value']), round(d['ms_per_step'],4))" | tee -a $S
done
