#!/bin/bash
# parity statistics of the in-record DIN / DLRM paths (end of round 3) next to the concatenating compositions
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
S=$OUT/summary_r03par.txt; : > $S
for C in c4_din c5_dlrm; do
  timeout 300 python scripts/parity_sweep.py --case $C --dists powerlaw --seeds 1 2 3 4 5 6 7 8 --variants default,no_inplace --par 16 \
      --out $OUT/parity_sweep_${C}_inplace.jsonl > $OUT/parity_sweep_${C}_inplace.log 2>&1
  echo "== $C exit $?" | tee -a $S
  tail -8 $OUT/parity_sweep_${C}_inplace.log | tee -a $S
done
