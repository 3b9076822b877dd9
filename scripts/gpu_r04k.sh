#!/bin/bash
# round 4, visit k: the last two k-tiles of a K loop without loads / masks (FX_GEMM_EDGE_PLAIN=2) — lab A/B (-1 = every body masked), tests, step A/B
TAG=${1:-r04k}
OUT=$PWD/gpurun_out; mkdir -p $OUT
S=$OUT/gemm_tail_bodies_$TAG.txt; : > $S
L=scripts/ubench/gemm_lab
for E in -1 1 2; do
  for SU in tower pairs; do
    echo "--- FX_GEMM_EDGE_PLAIN=$E suite $SU" | tee -a $S
    FX_GEMM_EDGE_PLAIN=$E FX_LAB_TAG=" [edge_plain=$E]" timeout 200 $L $SU 2>&1 | grep -v "^$" | cut -c1-200 | tee -a $S
  done
done
for SU in odd tower cross; do
echo "--- FX_GEMM_EDGE_PLAIN=2 suite $SU --check" | tee -a $S
FX_GEMM_EDGE_PLAIN=2 timeout 200 $L $SU --check 2>&1 | grep -v "^$" | cut -c1-200 | tee -a $S
done
echo "== GEMM / tower / cross tests, FX_GEMM_EDGE_PLAIN=2" | tee -a $S
FX_GEMM_EDGE_PLAIN=2 timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider -k "gemm or mlp or cross or tower or linear or multi or pair" 2>&1 | tail -4 | tee -a $S
echo "== step A/B (median step_us, value)" | tee -a $S
for R in 1 2; do for E in 1 2; do for M in DeepFM DCNv2 DIN DLRM xDeepFM; do
  FX_GEMM_EDGE_PLAIN=$E timeout 400 python bench.py --model $M --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-dcnv2 2>/dev/null | head -1 > $OUT/ab_tmp.json
  python -c "import json; d=json.load(open('$OUT/ab_tmp.json')); print('$M', 'edge_plain=$E', round(d['value']), round(d['ms_per_step'],4), d['step_us']['median'])" 2>&1 | tail -1 | tee -a $S
done; done; done
