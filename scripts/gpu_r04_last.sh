#!/bin/bash
# round 4, closing seconds: DCNv2 goldens + step after the ReLU-note identity check
OUT=$PWD/gpurun_out; mkdir -p $OUT
S=$OUT/last_r04.txt; : > $S
timeout 60 python -m pytest tests/test_gpu_models.py -m gpu -q --timeout 60 -p no:cacheprovider -k "dcnv2" 2>&1 | tail -2 | tee -a $S
timeout 60 python bench.py --model DCNv2 --steps 50 --warmup 10 --no-cpu-baseline --no-step-events 2>/dev/null | head -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('DCNv2', round(d['value']), round(d['ms_per_step'],4), 'kernel_sum_us', d.get('kernel_sum_us'), 'launches(ops)', d.get('kernel_sum_launches'))" | tee -a $S
