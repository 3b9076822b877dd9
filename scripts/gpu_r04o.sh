#!/bin/bash
# round 4, visit o2: tracebacks of the trajectory tests
TAG=${1:-r04o}
OUT=$PWD/gpurun_out; mkdir -p $OUT
S=$OUT/bisect2_$TAG.txt; : > $S
for E in "FX_NOOP=1" "FX_HEAD_FUSED=0"; do
  echo "--- $E" | tee -a $S
  env $E timeout 900 python -m pytest "tests/test_gpu_models.py::test_training_trajectory_matches_reference" -m gpu -q -x --timeout 600 -p no:cacheprovider -k "deepfm_bn or deepfm_adam" 2>&1 | tail -60 | cut -c1-220 | tee -a $S
done
