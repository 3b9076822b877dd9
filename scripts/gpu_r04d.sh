#!/bin/bash
# round 4, visit d: first-step gradient parity on the same ReLU decisions (c3 seeds, A/B of the summation orders, c2, uniform)
TAG=${1:-r04d}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
S=$OUT/summary_$TAG.txt
rm -f $OUT/grad_parity_$TAG.jsonl
FX_GRAD_PARITY_REPORT=$OUT/grad_parity_$TAG.jsonl timeout 2400 python -m pytest tests/test_gpu_grad_parity.py -q --timeout 1500 -p no:cacheprovider -s > $OUT/pytest_grad_$TAG.log 2>&1
echo "pytest exit $?" | tee $S
grep "grad parity\]" $OUT/pytest_grad_$TAG.log | cut -c1-700 | tee -a $S
tail -5 $OUT/pytest_grad_$TAG.log | cut -c1-600 | tee -a $S
python scripts/grad_parity_table.py $OUT/grad_parity_$TAG.jsonl > $OUT/grad_parity_table_$TAG.txt 2>&1
