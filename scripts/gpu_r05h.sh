#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r05_gemm_x6_bias.txt; : > $O
for x in 1 0; do
  echo "== FX_GEMM_BF16X6=$x (column fx_gemm_f32 = production path)" >> $O
  FX_GEMM_BF16X6=$x timeout 120 ./scripts/ubench/gemm_x6s_lab fuxictr_amd/libfxctr.so bias >> $O 2>&1
done
grep -E "==|signed|relL2" $O | sed -e 's/.*| relL2/   relL2/' | cut -c1-170
