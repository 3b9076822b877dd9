#!/bin/bash
# round 4, visit f: the TRUE drop-in timed (the reference's own model_zoo classes behind patch.install(), a reference
# checkout staged at .ref_checkout for this one call), interleaved with the mirrors; the bf16x6 lab, second cut.
TAG=${1:-r04f}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
S=$OUT/summary_$TAG.txt
echo "== bf16x6 lab" | tee $S
timeout 300 scripts/ubench/gemm_bf16x6_lab 2>&1 | tee $OUT/gemm_bf16x6_lab_$TAG.txt | cut -c1-260 | tee -a $S
export FX_REFERENCE_ROOT=$PWD/.ref_checkout
echo "== the reference's own model_zoo classes on the HIP kernels (parity)" | tee -a $S
timeout 900 python -m pytest tests/test_dropin_reference_zoo.py -m gpu -q --timeout 600 -p no:cacheprovider > $OUT/pytest_zoo_$TAG.log 2>&1
echo "pytest exit $?" | tee -a $S; tail -3 $OUT/pytest_zoo_$TAG.log | tee -a $S
echo "== drop-in timing: reference model_zoo classes vs fuxictr_amd.zoo mirrors, interleaved" | tee -a $S
for M in DeepFM DCNv2 DIN DLRM xDeepFM; do for R in 1 2; do for Z in reference native; do
  timeout 400 python bench.py --model $M --zoo $Z --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-dcnv2 2>$OUT/zoo_${M}_${Z}_$TAG.err | head -1 > $OUT/bench_zoo_${M}_${Z}_$TAG.json
  python -c "import json; d=json.load(open('$OUT/bench_zoo_${M}_${Z}_$TAG.json')); print('$M', '$Z', round(d['value']), round(d['ms_per_step'],4), d['step_us']['median'], d['config']['model_classes'][:40])" 2>&1 | tail -1 | tee -a $S
done; done; done
