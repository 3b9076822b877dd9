#!/bin/bash
# last check of a round's HEAD: whole GPU suite, smoke(), the lines of the models the last change touched
TAG=${1:-last}
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
S=$OUT/${TAG}_last_check_summary.txt; : > $S
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -rs > $OUT/pytest_gpu_$TAG.log 2>&1
echo "pytest exit $?" | tee -a $S
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_gpu_$TAG.log | tail -20 | tee -a $S
grep -E "^E  " $OUT/pytest_gpu_$TAG.log | head -20 | cut -c1-300 | tee -a $S
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a $S
for M in DIN DLRM; do
  timeout 600 python bench.py --model $M --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing > $OUT/${TAG}_bench_${M}.json 2>> $OUT/bench_$TAG.err
  python -c "import json; d=json.load(open('$OUT/${TAG}_bench_${M}.json')); print('$M', round(d['value']), 'samples/s', round(d['ms_per_step'],4), 'ms steady (young', d.get('young_run',{}).get('ms_per_step'), ')')" | tee -a $S
done
