#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 80 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 80 -p no:cacheprovider -k "early_exit" 2>&1 | tail -12 | cut -c1-220 | tee $OUT/last2_r04.txt
