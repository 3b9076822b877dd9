#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r05_gemm_x6w_lab_${1:-a}.txt; : > $O
FX_GEMM_BF16X6=1 timeout 200 ./scripts/ubench/gemm_x6w_lab fuxictr_amd/libfxctr.so ${2:-quick} >> $O 2>&1
echo "exit $?" >> $O
grep -E "us |exit|elements" $O | sed -e 's/ta[01] tb[01] sk[0-9] epi[01] //' -e 's/| max.max.*//' | cut -c1-230
