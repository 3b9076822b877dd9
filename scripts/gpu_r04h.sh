#!/bin/bash
# round 4, visit h: the tower's first layer (624 = 39 x 16 wide) under forced launch configurations — as it is and padded to 640
TAG=${1:-r04h}
OUT=$PWD/gpurun_out; mkdir -p $OUT
S=$OUT/gemm_first_layer_$TAG.txt; : > $S
L=scripts/ubench/gemm_lab
run() { echo "--- $*" | tee -a $S; env "$@" FX_LAB_TAG=" [$*]" timeout 120 $L first 2>&1 | grep -v "^$" | cut -c1-200 | tee -a $S; }
run FX_NOOP=1
run FX_GEMM_MULTI=0
for CFG in "0,4;0" "1,4;1" "0,8;0" "1,8;1" "0,2;0" "1,2;1" "0,4;1" "1,4;0" "0,8;1" "0,6;1" "1,6;1" "0,3;0" "1,3;1" "0,5;1" "0,1;0"; do
  run FX_MULTI_CFG="$CFG"
done
echo "== bf16x6 lab, third cut (8 waves, two LDS stages)" | tee -a $S
timeout 300 scripts/ubench/gemm_bf16x6_lab 2>&1 | tee $OUT/gemm_bf16x6_lab_$TAG.txt | cut -c1-260 | tee -a $S
export FX_REFERENCE_ROOT=$PWD/.ref_checkout
echo "== drop-in timing, DCNv2 (stock nn.Linear head on the native GEMM)" | tee -a $S
for R in 1 2; do for Z in reference native; do
  timeout 400 python bench.py --model DCNv2 --zoo $Z --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-dcnv2 2>/dev/null | head -1 > $OUT/ab_tmp.json
  python -c "import json; d=json.load(open('$OUT/ab_tmp.json')); print('DCNv2', '$Z', round(d['value']), round(d['ms_per_step'],4), d['step_us']['median'])" 2>&1 | tail -1 | tee -a $S
done; done
timeout 600 python -m pytest tests/test_dropin_reference_zoo.py -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -2 | tee -a $S
