#!/bin/bash
TAG=${1:-r03h}
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
S=$OUT/summary_$TAG.txt
echo "== pytest fused / kernels" | tee $S
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_kernels.py tests/test_gpu_models.py -m gpu -q -x --timeout 600 -p no:cacheprovider > $OUT/pytest_subset_$TAG.log 2>&1
echo "pytest exit $?" | tee -a $S
tail -4 $OUT/pytest_subset_$TAG.log | tee -a $S
echo "== rocprofv3 timeline DeepFM" | tee -a $S
rm -rf /tmp/prof_$TAG
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- \
    python $REPO/bench.py --no-cpu-baseline --no-kernel-timing --no-dcnv2 > $OUT/prof_bench_$TAG.json 2> $OUT/prof_$TAG.err)
TR=$(find /tmp/prof_$TAG -name '*kernel_trace.csv' | head -1)
python scripts/step_timeline.py $TR 3 > $OUT/timeline_deepfm_$TAG.txt; cat $OUT/timeline_deepfm_$TAG.txt | tee -a $S
echo "== bench" | tee -a $S
timeout 900 python bench.py --no-cpu-baseline > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
python - <<PY | tee -a $S
import json
d = json.loads(open("$OUT/bench_$TAG.json").readline())
print("value", round(d["value"]), "ms", round(d["ms_per_step"], 4), "roofline", round(d["roofline"]["frac"], 3))
print("sparse", d["roofline_sparse"]["us_per_step"], d["roofline_sparse"].get("by_launch_us"))
print("dcnv2", round(d["dcnv2"]["value"]), round(d["dcnv2"]["ms_per_step"], 4), round(d["dcnv2"]["roofline"]["frac"], 3))
for k, v in d["roofline"]["by_shape_MxNxK"].items(): print("  ", k, v)
for k, v in d["dcnv2"]["roofline"]["by_shape_MxNxK"].items(): print("  dcnv2", k, v)
PY
