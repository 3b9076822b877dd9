#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r05_gemm_prod_variants.txt; : > $O
for so in fuxictr_amd/libfxctr.so scripts/ubench/libfx_ONE_CHAIN.so scripts/ubench/libfx_NO_MASK.so scripts/ubench/libfx_ONE_CHAINDX6_NO_MASK.so; do
  echo "== $so" >> $O
  FX_GEMM_BF16X6=1 timeout 120 ./scripts/ubench/gemm_x6s_lab $so quick >> $O 2>&1
done
grep -E "==|us " $O | sed -e 's/ta[01] tb[01] sk[0-9] epi[01] //' -e 's/| relL2.*//' | cut -c1-200
