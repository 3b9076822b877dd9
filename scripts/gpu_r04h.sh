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
