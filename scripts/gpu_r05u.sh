#!/bin/bash
# the TRUE drop-in (the reference's own model_zoo classes behind patch.install(); checkout staged next to the repo
# for this call only, removed afterwards) against the mirrors, interleaved on one box; the drop-in GPU tests; the
# end-to-end rates (host tensors per step / DeviceNpzDataLoader)
mkdir -p gpurun_out; export TMPDIR=/tmp
export FX_REFERENCE_ROOT=$PWD/.ref_checkout
S=gpurun_out/r05_dropin_timing.txt; : > $S
timeout 900 python -m pytest tests/test_dropin_reference_zoo.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3 | tee -a $S
for R in 1 2; do for M in DeepFM DCNv2 DIN; do for Z in native reference; do
  timeout 400 python bench.py --model $M --zoo $Z --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-dcnv2 --no-parity --no-uniform 2>/dev/null | head -1 > gpurun_out/ab_tmp.json
  python -c "import json; d=json.load(open('gpurun_out/ab_tmp.json')); print('$M', 'zoo=$Z', 'run $R', round(d['value']), 'samples/s', round(d['ms_per_step'],4), 'ms (steady state after', d['warmup'], 'steps)')" 2>&1 | tail -1 | tee -a $S
done; done; done
for F in --host-inputs --loader; do
  timeout 400 python bench.py $F --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-dcnv2 --no-parity --no-uniform 2>/dev/null | head -1 > gpurun_out/ab_tmp.json
  python -c "import json; d=json.load(open('gpurun_out/ab_tmp.json')); print('DeepFM $F', round(d['value']), 'samples/s', round(d['ms_per_step'],4), 'ms;', d['config']['inputs'][:90])" 2>&1 | tail -1 | tee -a $S
done
