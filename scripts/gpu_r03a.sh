#!/bin/bash
# round 3, visit a: GEMM lab (tiles x TR epilogue, timelines, SQ/GRBM counters), c5 DLRM multi-seed parity
# sweep with A/B switches, a GPU test subset (TR epilogue, bf16 generic path), the reference's own zoo
# classes on the real kernels (needs FX_REFERENCE_ROOT), the default bench line.
TAG=${1:-r03a}
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
S=$OUT/summary_$TAG.txt
LAB=$REPO/scripts/ubench/gemm_lab
echo "== gemm lab" | tee $S
L=$OUT/gemm_lab_$TAG.txt; : > $L
for TILE in 64x64 128x64 128x128; do
  for TR in 0 1; do
    FX_GEMM_TILE=$TILE FX_GEMM_TR=$TR timeout 120 $LAB all >> $L 2>&1
  done
done
FX_GEMM_TR=1 timeout 120 $LAB all >> $L 2>&1          # tile auto
FX_GEMM_TR=0 timeout 120 $LAB all >> $L 2>&1
echo "-- correctness (TR on, every tile; bit-identical to a k-ordered fmaf chain when split_k = 1)" >> $L
for TILE in 64x64 128x64 128x128; do
  FX_GEMM_TILE=$TILE timeout 300 $LAB odd --check >> $L 2>&1
done
timeout 300 $LAB tower --check >> $L 2>&1
timeout 300 $LAB cross --check >> $L 2>&1
echo "-- timelines" >> $L
for TILE in 64x64 128x64 128x128; do
  for TR in 0 1; do
    FX_GEMM_TILE=$TILE FX_GEMM_TR=$TR timeout 120 $LAB tower --trace >> $L 2>&1
  done
done
FX_GEMM_TR=1 timeout 120 $LAB cross --trace >> $L 2>&1
FX_GEMM_TILE=64x64 FX_GEMM_TR=1 timeout 120 $LAB ksweep --trace >> $L 2>&1
grep -c MISMATCH $L | sed 's/^/MISMATCH lines: /' | tee -a $S
grep "tile=auto" $L | head -24 | tee -a $S
echo "== lab SQ / GRBM counters" | tee -a $S
CTRS="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
for TILE in 64x64 128x64 128x128; do
  rm -rf /tmp/pmc_lab_$TILE
  (cd /tmp && FX_GEMM_TILE=$TILE timeout 300 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d /tmp/pmc_lab_$TILE -- $LAB all > /dev/null 2> $OUT/pmc_lab_${TILE}_$TAG.err)
  python scripts/pmc_lab.py "$(find /tmp/pmc_lab_$TILE -name '*counter_collection.csv' | head -1)" \
      "$(find /tmp/pmc_lab_$TILE -name '*kernel_trace.csv' | head -1)" 2>&1 | tee -a $OUT/pmc_lab_$TAG.txt | tail -12 | tee -a $S
done
echo "== parity sweep c5_dlrm" | tee -a $S
timeout 1200 python scripts/parity_sweep.py --case c5_dlrm --seeds 1 2 3 4 5 6 7 8 \
    --variants default,dot_valu,no_pad,no_pair,splitk1 --par 16 --out $OUT/parity_sweep_c5_$TAG.jsonl > $OUT/parity_sweep_c5_$TAG.log 2>&1
tail -30 $OUT/parity_sweep_c5_$TAG.log | tee -a $S
echo "== pytest subset" | tee -a $S
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bf16.py tests/test_gpu_models.py tests/test_c1_tiny_npz.py -m gpu -q -x --timeout 600 -p no:cacheprovider > $OUT/pytest_subset_$TAG.log 2>&1
echo "pytest exit $?" | tee -a $S
tail -5 $OUT/pytest_subset_$TAG.log | tee -a $S
if [ -n "$FX_REFERENCE_ROOT" ] && [ -d "$FX_REFERENCE_ROOT/fuxictr" ]; then
  echo "== reference zoo classes on the HIP kernels (patch.install(); reference checkout at FX_REFERENCE_ROOT)" | tee -a $S
  timeout 600 python -m pytest tests/test_dropin_reference_zoo.py -m gpu -q -rs --timeout 300 -p no:cacheprovider -v > $OUT/pytest_dropin_gpu_$TAG.log 2>&1
  echo "dropin exit $?" | tee -a $S
  grep -E "PASSED|FAILED|SKIPPED|passed|failed" $OUT/pytest_dropin_gpu_$TAG.log | tail -12 | tee -a $S
fi
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
