#!/bin/bash
# round 4, closing seconds: the tests that run fit() on the GPU, after the loss window moved to the device
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 45 python -m pytest tests/test_c1_tiny_npz.py tests/test_dataloader.py "tests/test_gpu_models.py::test_fit_evaluate_checkpoint_roundtrip" -m gpu -q --timeout 45 -p no:cacheprovider 2>&1 | tail -3 | cut -c1-200 | tee $OUT/fit_r04.txt
