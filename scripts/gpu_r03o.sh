#!/bin/bash
# DCNv2: cross layer + deep layer of a depth as ONE forward grid of 64x64 tiles (FX_GEMM_FWDPAIR)
TAG=${1:-r03u}
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
S=$OUT/summary_$TAG.txt
echo "== pytest" | tee $S
timeout 1200 python -m pytest tests/test_gpu_models.py tests/test_gpu_kernels.py -m gpu -q -x --timeout 900 -p no:cacheprovider -k "dcn or DCN or gemm or cross" > $OUT/pytest_$TAG.log 2>&1
echo "pytest exit $?" | tee -a $S
tail -3 $OUT/pytest_$TAG.log | tee -a $S
echo "== A/B fwd pair (DCNv2)" | tee -a $S
bash scripts/gpu_ab.sh fwdpair_$TAG DCNv2 "FX_GEMM_FWDPAIR=1" "FX_GEMM_FWDPAIR=0" 3 | tee -a $S
M=DCNv2
rm -rf /tmp/prof_${TAG}_$M
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_$M -- \
    python $REPO/bench.py --model $M --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > /dev/null 2> $OUT/prof_${TAG}_$M.err)
TR=$(find /tmp/prof_${TAG}_$M -name '*kernel_trace.csv' | head -1)
python scripts/step_timeline.py $TR 3 > $OUT/timeline_${M}_$TAG.txt
head -16 $OUT/timeline_${M}_$TAG.txt | tee -a $S; tail -1 $OUT/timeline_${M}_$TAG.txt | tee -a $S
