#!/bin/bash
# the whole GPU suite, as the driver runs it (plus the durations of the slowest tests)
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-full}
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider --durations=15 > gpurun_out/pytest_$TAG.log 2>&1
echo "pytest exit $?" | tee gpurun_out/pytest_${TAG}_summary.txt
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest_$TAG.log | tail -40 | tee -a gpurun_out/pytest_${TAG}_summary.txt
grep -E "^E  " gpurun_out/pytest_$TAG.log | head -40 | cut -c1-300 | tee -a gpurun_out/pytest_${TAG}_summary.txt
grep -A18 "slowest" gpurun_out/pytest_$TAG.log | tee -a gpurun_out/pytest_${TAG}_summary.txt
