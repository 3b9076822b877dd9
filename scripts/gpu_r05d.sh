#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r05_gemm_x6s_lab_d.txt
timeout 120 ./scripts/ubench/gemm_x6s_lab fuxictr_amd/libfxctr.so quick > $O 2>&1
grep -E "x6s|elements" $O | cut -c1-230
