#!/bin/bash
# DLRM with the bottom tower's vector in the gather record: parity tests, same-box A/B, timeline
TAG=${1:-r03w}
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
S=$OUT/summary_$TAG.txt
echo "== pytest dlrm" | tee $S
timeout 1200 python -m pytest tests/test_gpu_models.py tests/test_gpu_kernels.py tests/test_gpu_dist.py -m gpu -q -x -k "dlrm or DLRM or dot_interaction" --timeout 900 -p no:cacheprovider > $OUT/pytest_$TAG.log 2>&1
echo "pytest exit $?" | tee -a $S
tail -5 $OUT/pytest_$TAG.log | tee -a $S
echo "== A/B in-record (DLRM)" | tee -a $S
bash scripts/gpu_ab.sh dlrm_$TAG DLRM "FX_DLRM_INPLACE=1" "FX_DLRM_INPLACE=0" 2 | tee -a $S
M=DLRM
rm -rf /tmp/prof_${TAG}_$M
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_$M -- \
    python $REPO/bench.py --model $M --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > /dev/null 2> $OUT/prof_${TAG}_$M.err)
TR=$(find /tmp/prof_${TAG}_$M -name '*kernel_trace.csv' | head -1)
python scripts/step_timeline.py $TR 3 > $OUT/timeline_${M}_$TAG.txt
cat $OUT/timeline_${M}_$TAG.txt | tee -a $S
