#!/bin/bash
# the baseline-shape parity sweep (8 model seeds, both id distributions) on the current kernels,
# native default + kernel switches off (VARIANTS=...), yardsticks refreshed.  usage: gpu_parity_sweep.sh TAG
TAG=${1:-r05}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for C in c2_deepfm c3_dcnv2 c4_din c5_dlrm; do
  T0=$(date +%s)
  timeout 900 python scripts/parity_sweep.py --case $C --variants ${VARIANTS:-default,x6_off,series_off} --par 16 --out $OUT/${TAG}_parity_sweep_$C.jsonl > $OUT/${TAG}_parity_sweep_${C}_rms.txt 2>&1
  echo "== $C exit $? ($(( $(date +%s) - T0 )) s)"; cat $OUT/${TAG}_parity_sweep_${C}_rms.txt | tail -12
done
python scripts/make_parity_yardsticks.py $OUT/${TAG}_parity_yardsticks.json $OUT/${TAG}_parity_sweep_c*.jsonl
python scripts/parity_sweep_summary.py $OUT/${TAG}_parity_sweep_c*.jsonl > $OUT/${TAG}_parity_sweep_summary.txt; cat $OUT/${TAG}_parity_sweep_summary.txt
