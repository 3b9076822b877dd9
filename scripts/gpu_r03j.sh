#!/bin/bash
# CIN on the matrix cores: kernel parity, xDeepFM goldens, same-box A/B against the VALU kernels, timeline
TAG=${1:-r03j}
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
S=$OUT/summary_$TAG.txt
echo "== pytest cin" | tee $S
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "cin" --timeout 600 -p no:cacheprovider > $OUT/pytest_cin_$TAG.log 2>&1
echo "pytest exit $?" | tee -a $S
tail -8 $OUT/pytest_cin_$TAG.log | tee -a $S
timeout 900 python -m pytest tests/test_gpu_models.py -m gpu -q -x -k "xdeepfm or xDeepFM" --timeout 600 -p no:cacheprovider > $OUT/pytest_xdeepfm_$TAG.log 2>&1
echo "pytest xdeepfm exit $?" | tee -a $S
tail -4 $OUT/pytest_xdeepfm_$TAG.log | tee -a $S
echo "== A/B CIN MFMA (xDeepFM)" | tee -a $S
bash scripts/gpu_ab.sh cin_$TAG xDeepFM "FX_CIN_MFMA=1" "FX_CIN_MFMA=0" 2 | tee -a $S
M=xDeepFM
rm -rf /tmp/prof_${TAG}_$M
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_$M -- \
    python $REPO/bench.py --model $M --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-dcnv2 > /dev/null 2> $OUT/prof_${TAG}_$M.err)
TR=$(find /tmp/prof_${TAG}_$M -name '*kernel_trace.csv' | head -1)
python scripts/step_timeline.py $TR 3 > $OUT/timeline_${M}_$TAG.txt; echo "$M $(tail -1 $OUT/timeline_${M}_$TAG.txt)" | tee -a $S
ST=$(find /tmp/prof_${TAG}_$M -name '*kernel_stats.csv' | head -1); cp $ST $OUT/kernel_stats_${M}_$TAG.csv
cat $OUT/timeline_${M}_$TAG.txt | tee -a $S
