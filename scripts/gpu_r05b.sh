#!/bin/bash
# round 5 visit b: ablations of the in-kernel-split kernel (what bounds a k tile?)
mkdir -p gpurun_out
O=gpurun_out/r05_gemm_x6s_lab_b.txt
: > $O
for v in "" -NO_MFMA -NO_SPLIT -NO_GLOAD -NO_LDSW; do
  echo "== variant '$v'" >> $O
  timeout 120 ./scripts/ubench/gemm_x6s_lab$v nolib quick >> $O 2>&1
done
grep -E "variant|x6s" $O | cut -c1-150
