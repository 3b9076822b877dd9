#!/bin/bash
# round 4, last call: the reference's own model_zoo classes vs the mirrors after the one-pass head (checkout staged at .ref_checkout for this call only)
OUT=$PWD/gpurun_out; mkdir -p $OUT
export FX_REFERENCE_ROOT=$PWD/.ref_checkout
S=$OUT/dropin_timing_final_r04.txt; : > $S
for M in DeepFM DCNv2 DIN; do for Z in reference native; do
  timeout 60 python bench.py --model $M --zoo $Z --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-dcnv2 --no-step-events 2>/dev/null | head -1 > $OUT/ab_tmp.json
  python -c "import json; d=json.load(open('$OUT/ab_tmp.json')); print('$M', '$Z', round(d['value']), round(d['ms_per_step'],4))" 2>&1 | tail -1 | tee -a $S
done; done
