#!/bin/bash
# round 4, closing check of the 16-step replay window: exact-mode tests, default-shape bench line, steady state
TAG=${1:-r04final5}
OUT=$PWD/gpurun_out; mkdir -p $OUT
S=$OUT/summary_$TAG.txt
timeout 200 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 200 -p no:cacheprovider -k "exact_mode or catchup or adam" 2>&1 | tail -2 | tee $S
timeout 200 python bench.py --no-cpu-baseline --no-kernel-timing --no-dcnv2 2>/dev/null | head -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('default shape:', round(d['value']), d['ms_per_step'], d['step_us']['median'])" | tee -a $S
timeout 200 python bench.py --steps 50 --warmup 300 --no-cpu-baseline --no-kernel-timing --no-dcnv2 2>/dev/null | head -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('warm-up 300:', round(d['value']), d['ms_per_step'], d['step_us']['median'])" | tee -a $S
