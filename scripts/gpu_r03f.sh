#!/bin/bash
# round 3, visit f: forced tile / split configurations of the DCNv2 grids (planner check)
TAG=${1:-r03f}
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
LAB=$REPO/scripts/ubench/gemm_lab
L=$OUT/gemm_lab_$TAG.txt; : > $L
FX_MULTI_VERBOSE=1 FX_LAB_TAG=" planner" timeout 200 $LAB dcn >> $L 2>&1
# forward pairs: problems = cross, deep
for CFG in "0;0" "1;0" "1;1" "0;1"; do
  FX_MULTI_CFG="$CFG" FX_LAB_TAG=" cfg=$CFG" timeout 200 $LAB dcn 2>&1 | grep "cross fwd + deep" >> $L
done
# backward: problems = cross dW, cross dX, deep dW, deep dX
for CFG in "0,4;0;0,4;0" "1,4;1;0,4;0" "1,8;1;0,4;0" "1,4;1;0,8;0" "0,8;0;0,4;0" "1,2;1;0,4;0" "1,4;1;0,3;0" "1,4;1;0,6;0" "1,6;1;0,4;0" "1,3;1;0,4;0" "0,4;1;0,4;0" "1,4;0;0,4;0"; do
  FX_MULTI_CFG="$CFG" FX_LAB_TAG=" cfg=$CFG" timeout 200 $LAB dcn 2>&1 | grep "cross pair + deep" >> $L
done
FX_GEMM_MULTI=0 FX_LAB_TAG=" multi=0" timeout 200 $LAB dcn >> $L 2>&1
cat $L
