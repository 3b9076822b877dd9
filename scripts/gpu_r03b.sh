#!/bin/bash
# round 3, visit b: GEMM lab after the epilogue restructure + the multi-problem grid (pairs on 128-row
# tiles, planner vs forced configurations), yardstick sweeps for c2/c3/c4, GPU test subset, bench.
TAG=${1:-r03b}
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
S=$OUT/summary_$TAG.txt
LAB=$REPO/scripts/ubench/gemm_lab
L=$OUT/gemm_lab_$TAG.txt; : > $L
echo "== gemm lab" | tee $S
echo "-- single GEMMs, new epilogue, per tile" >> $L
for TILE in 64x64 128x64 128x128; do
  FX_GEMM_TILE=$TILE timeout 120 $LAB tower >> $L 2>&1
  FX_GEMM_TILE=$TILE timeout 120 $LAB cross >> $L 2>&1
done
timeout 120 $LAB all >> $L 2>&1
echo "-- two workgroups per CU" >> $L
FX_GEMM_TILE=128x128 timeout 120 $LAB two >> $L 2>&1
FX_GEMM_TILE=128x64 timeout 120 $LAB two >> $L 2>&1
echo "-- pairs: planner" >> $L
timeout 120 $LAB pairs --check >> $L 2>&1
echo "-- pairs: 64x64 pair kernel of round 2 (FX_GEMM_MULTI=0)" >> $L
FX_GEMM_MULTI=0 FX_LAB_TAG=" multi=0" timeout 120 $LAB pairs >> $L 2>&1
echo "-- pairs: forced configurations (tile,sk of dW ; tile of dX; tile 0 = 128x128, 1 = 128x64)" >> $L
for CFG in "0,4;0" "0,8;0" "0,2;0" "0,3;0" "0,6;0" "1,4;1" "1,8;1" "1,2;1" "0,4;1" "1,4;0" "1,3;1" "1,6;1" "0,5;1"; do
  FX_MULTI_CFG="$CFG" FX_LAB_TAG=" cfg=$CFG" timeout 120 $LAB pairs >> $L 2>&1
done
echo "-- correctness" >> $L
for TILE in 64x64 128x64 128x128; do
  FX_GEMM_TILE=$TILE timeout 300 $LAB odd --check >> $L 2>&1
done
timeout 300 $LAB tower --check >> $L 2>&1
timeout 300 $LAB cross --check >> $L 2>&1
echo "-- timelines" >> $L
FX_GEMM_TILE=128x128 timeout 120 $LAB tower --trace >> $L 2>&1
FX_GEMM_TILE=128x128 timeout 120 $LAB two --trace >> $L 2>&1
grep -c MISMATCH $L | sed 's/^/MISMATCH lines: /' | tee -a $S
grep "tile=auto tr=1\]" $L | head -40 | tee -a $S
echo "== parity yardstick sweeps (8 seeds)" | tee -a $S
for CASE in c2_deepfm c4_din c3_dcnv2; do
  timeout 900 python scripts/parity_sweep.py --case $CASE --seeds 1 2 3 4 5 6 7 8 --variants default --par 16 \
      --out $OUT/parity_sweep_${CASE}_$TAG.jsonl > $OUT/parity_sweep_${CASE}_$TAG.log 2>&1
  tail -8 $OUT/parity_sweep_${CASE}_$TAG.log | tee -a $S
done
echo "== pytest subset" | tee -a $S
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bf16.py tests/test_gpu_models.py -m gpu -q -x --timeout 600 -p no:cacheprovider > $OUT/pytest_subset_$TAG.log 2>&1
echo "pytest exit $?" | tee -a $S
tail -5 $OUT/pytest_subset_$TAG.log | tee -a $S
echo "== bench (default command)" | tee -a $S
timeout 900 python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
echo "bench exit $?" | tee -a $S
python - <<PY | tee -a $S
import json
d = json.loads(open("$OUT/bench_$TAG.json").readline())
print("value", round(d["value"]), "ms", round(d["ms_per_step"], 4), "roofline", round(d["roofline"]["frac"], 3))
for k in ("roofline_sparse", "roofline_gather", "roofline_gather_b32768"):
    if k in d:
        print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in d[k].items() if a in ("frac", "us_per_step", "avg_launch_us", "achieved", "by_launch_us", "distinct_batches_replayed")})
print("dcnv2", round(d["dcnv2"]["value"]), round(d["dcnv2"]["ms_per_step"], 4), round(d["dcnv2"]["roofline"]["frac"], 3))
for k, v in d["roofline"]["by_shape_MxNxK"].items(): print("  ", k, v)
for k, v in d["dcnv2"]["roofline"]["by_shape_MxNxK"].items(): print("  dcnv2", k, v)
PY
for M in DIN DLRM xDeepFM; do
  timeout 300 python bench.py --model $M --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing > $OUT/bench_${M}_$TAG.json 2>/dev/null
  python -c "import json; d=json.load(open('$OUT/bench_${M}_$TAG.json')); print('$M', round(d['value']), round(d['ms_per_step'],4))" | tee -a $S
done
