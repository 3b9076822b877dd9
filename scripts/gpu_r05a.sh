#!/bin/bash
# round 5 visit a: the in-kernel-split bf16x6 GEMM lab against the production fp32-MFMA kernel (same box)
mkdir -p gpurun_out
timeout 300 ./scripts/ubench/gemm_x6s_lab fuxictr_amd/libfxctr.so > gpurun_out/r05_gemm_x6s_lab_a.txt 2>&1
echo "exit $?" >> gpurun_out/r05_gemm_x6s_lab_a.txt
tail -40 gpurun_out/r05_gemm_x6s_lab_a.txt
