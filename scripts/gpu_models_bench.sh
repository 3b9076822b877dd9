#!/bin/bash
# bench lines of the other BASELINE models (no cpu baseline, no DCNv2 sub-run) + a DIN step timeline
TAG=$1
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for M in ${MODELS:-DIN DLRM xDeepFM}; do
  timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-dcnv2 --model $M > $OUT/bench_${TAG}_$M.json 2> $OUT/bench_${TAG}_$M.err
  python - $OUT/bench_${TAG}_$M.json $M <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    sp = d.get("roofline_sparse", {})
    print("%s: %.0f samples/s  %.4f ms/step  gemm %.1f us  sparse %.1f us/%s launches" % (
        sys.argv[2], d["value"], d["ms_per_step"], d.get("roofline", {}).get("gemm_us_per_step", 0),
        sp.get("us_per_step", 0), sp.get("launches_per_step")))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
  tail -2 $OUT/bench_${TAG}_$M.err
done
rm -rf /tmp/prof_din
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_din -- \
    python $REPO/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-dcnv2 --model DIN > /dev/null 2> $OUT/prof_din_$TAG.err)
TR=$(find /tmp/prof_din -name '*kernel_trace.csv' | head -1)
python scripts/step_timeline.py $TR 3 > $OUT/timeline_din_$TAG.txt; tail -1 $OUT/timeline_din_$TAG.txt
