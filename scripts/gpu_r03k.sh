#!/bin/bash
# DIN in-record attention: parity tests, same-box A/B against the concatenating composition, timeline
TAG=${1:-r03o}
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
S=$OUT/summary_$TAG.txt
echo "== pytest din" | tee $S
timeout 1200 python -m pytest tests/test_gpu_models.py tests/test_gpu_kernels.py tests/test_gpu_fused.py tests/test_gpu_bf16.py -m gpu -q -x -k "din or DIN or dice or attn" --timeout 900 -p no:cacheprovider > $OUT/pytest_din_$TAG.log 2>&1
echo "pytest exit $?" | tee -a $S
tail -6 $OUT/pytest_din_$TAG.log | tee -a $S
echo "== A/B in-record (DIN)" | tee -a $S
bash scripts/gpu_ab.sh din_$TAG DIN "FX_DIN_INPLACE=1" "FX_DIN_INPLACE=0" 2 | tee -a $S
M=DIN
rm -rf /tmp/prof_${TAG}_$M
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_$M -- \
    python $REPO/bench.py --model $M --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-dcnv2 > /dev/null 2> $OUT/prof_${TAG}_$M.err)
TR=$(find /tmp/prof_${TAG}_$M -name '*kernel_trace.csv' | head -1)
python scripts/step_timeline.py $TR 3 > $OUT/timeline_${M}_$TAG.txt
cat $OUT/timeline_${M}_$TAG.txt | tee -a $S
