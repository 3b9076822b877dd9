#!/bin/bash
# round 2, visit M: DIN attention q-split formulation — tests, A/B of the passes, bench, timeline
TAG=${1:-r02m}
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
S=$OUT/summary_$TAG.txt
echo "== pytest fused DIN attention + DIN models" | tee $S
timeout 900 python -m pytest tests/test_gpu_din_attn.py tests/test_gpu_models.py -m gpu -q -k "din or DIN" --timeout 600 -p no:cacheprovider > $OUT/pytest_dinattn_$TAG.log 2>&1
echo "pytest exit $?" | tee -a $S
tail -25 $OUT/pytest_dinattn_$TAG.log | tee -a $S
echo "== per-pass times" | tee -a $S
for Q in 1 0; do echo "FX_DIN_ATTN_QSPLIT=$Q" | tee -a $S; FX_DIN_ATTN_QSPLIT=$Q timeout 300 python scripts/din_attn_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $S; done
for W in 1024 1366 4096; do FX_DIN_ATTN_BWD_WAVES=$W FX_DIN_ATTN_FWD_WAVES=$W timeout 300 python scripts/din_attn_bench.py 2>&1 | grep -E "B=|apply|stats|sums" | tee -a $S; done
echo "== bench DIN" | tee -a $S
timeout 600 python bench.py --model DIN --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench_din_$TAG.json 2> $OUT/bench_din_$TAG.err
cut -c1-330 $OUT/bench_din_$TAG.json | tee -a $S
echo "== timeline DIN" | tee -a $S
rm -rf /tmp/prof_$TAG
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- \
    python $REPO/bench.py --model DIN --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > $OUT/prof_bench_$TAG.json 2> $OUT/prof_$TAG.err)
STATS=$(ls -t $(find /tmp/prof_$TAG -name '*kernel_stats.csv') 2>/dev/null | head -1)
if [ -n "$STATS" ]; then cp $STATS $OUT/kernel_stats_din_$TAG.csv; fi
TR=$(find /tmp/prof_$TAG -name '*kernel_trace.csv' | head -1)
python scripts/step_timeline.py $TR 3 > $OUT/timeline_din_$TAG.txt; grep -E "din_attn|da_|kernels" $OUT/timeline_din_$TAG.txt | cut -c1-110 | tee -a $S
