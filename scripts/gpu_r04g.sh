#!/bin/bash
# round 4, visit g: the whole GPU suite after the round's changes; the true drop-in timed again (DCNv2 / DIN hand-offs,
# DLRM input shapes), the reference's own classes on the HIP kernels (parity).
TAG=${1:-r04g}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
S=$OUT/summary_$TAG.txt
export FX_REFERENCE_ROOT=$PWD/.ref_checkout
echo "== pytest -m gpu (whole suite, reference checkout staged)" | tee $S
FX_PARITY_REPORT=$OUT/parity_$TAG.jsonl timeout 2400 python -m pytest tests -m gpu -q --timeout 1200 -p no:cacheprovider -rs > $OUT/pytest_gpu_$TAG.log 2>&1
echo "pytest exit $?" | tee -a $S
tail -12 $OUT/pytest_gpu_$TAG.log | cut -c1-300 | tee -a $S
echo "== drop-in timing: reference model_zoo classes vs fuxictr_amd.zoo mirrors, interleaved" | tee -a $S
for M in DCNv2 DIN DLRM; do for R in 1 2; do for Z in reference native; do
  timeout 400 python bench.py --model $M --zoo $Z --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-dcnv2 2>$OUT/zoo_${M}_${Z}_$TAG.err | head -1 > $OUT/bench_zoo_${M}_${Z}_$TAG.json
  python -c "import json; d=json.load(open('$OUT/bench_zoo_${M}_${Z}_$TAG.json')); print('$M', '$Z', round(d['value']), round(d['ms_per_step'],4), d['step_us']['median'], d['config']['model_classes'][:40])" 2>&1 | tail -1 | tee -a $S
done; done; done
