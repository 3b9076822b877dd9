#!/bin/bash
# production GEMM kernels (libfxctr.so, column "fp32-mfma" = fx_gemm_f32 under the given switch) on the step's shapes
mkdir -p gpurun_out
O=gpurun_out/r05_gemm_prod_${1:-f}.txt; : > $O
for x in 1 0; do
  echo "== FX_GEMM_BF16X6=$x" >> $O
  FX_GEMM_BF16X6=$x timeout 120 ./scripts/ubench/gemm_x6s_lab fuxictr_amd/libfxctr.so >> $O 2>&1
done
grep -E "==|x6s" $O | sed -e 's/relL2.*//' | cut -c1-170
