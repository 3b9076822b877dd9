#!/bin/bash
# A/B of environment switches on the bench, interleaved runs on ONE box.  usage: gpu_ab.sh TAG MODEL "ENV_A" "ENV_B" [reps]
TAG=$1; MODEL=$2; A=$3; B=$4; REPS=${5:-3}
OUT=$PWD/gpurun_out; mkdir -p $OUT
S=$OUT/ab_$TAG.txt; : > $S
for R in $(seq 1 $REPS); do
  for V in A B; do
    if [ $V = A ]; then E="$A"; else E="$B"; fi
    env $E timeout 300 python bench.py --model $MODEL --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-dcnv2 2>/dev/null | head -1 > $OUT/ab_tmp.json
    python -c "import json; d=json.load(open('$OUT/ab_tmp.json')); print('$V [$E]', round(d['value']), round(d['ms_per_step'],4))" | tee -a $S
  done
done
