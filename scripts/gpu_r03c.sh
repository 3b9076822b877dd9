#!/bin/bash
# round 3, visit c: why is the two-problem grid slower than its parts?  mix suite (same / different bodies
# co-resident) with timelines, rowsum dealt round-robin, pairs again.
TAG=${1:-r03c}
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
S=$OUT/summary_$TAG.txt
LAB=$REPO/scripts/ubench/gemm_lab
L=$OUT/gemm_lab_$TAG.txt; : > $L
echo "-- mix (planner)" >> $L
timeout 200 $LAB mix >> $L 2>&1
echo "-- mix forced 128x128, sk 4" >> $L
FX_MULTI_CFG="0,4;0,4" FX_LAB_TAG=" cfg=0,4;0,4" timeout 200 $LAB mix --trace >> $L 2>&1
echo "-- mix forced 128x64, sk 4" >> $L
FX_MULTI_CFG="1,4;1,4" FX_LAB_TAG=" cfg=1,4;1,4" timeout 200 $LAB mix >> $L 2>&1
echo "-- singles (rowsum dealt round-robin)" >> $L
for TILE in 64x64 128x128; do
  FX_GEMM_TILE=$TILE timeout 120 $LAB tower >> $L 2>&1
done
FX_GEMM_TILE=128x128 timeout 120 $LAB tower --trace >> $L 2>&1
echo "-- pairs" >> $L
timeout 120 $LAB pairs --check >> $L 2>&1
FX_GEMM_MULTI=0 FX_LAB_TAG=" multi=0" timeout 120 $LAB pairs --check >> $L 2>&1
echo "-- correctness" >> $L
for TILE in 64x64 128x64 128x128; do
  FX_GEMM_TILE=$TILE timeout 300 $LAB odd --check >> $L 2>&1
done
timeout 300 $LAB tower --check >> $L 2>&1
grep -c MISMATCH $L | sed 's/^/MISMATCH lines: /' | tee $S
grep "^\[" $L | head -70 | tee -a $S
