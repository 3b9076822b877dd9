#!/bin/bash
# sharded step on one RCCL rank: hipGraph segments vs the collectives recorded into the graph
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
S=$OUT/summary_r03s.txt; : > $S
run() {  # name timeout env...
  local name=$1 to=$2; shift 2
  echo "== $name [$*]" | tee -a $S
  env FX_SHARD_WORLD1=1 "$@" timeout -s KILL $to python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-dcnv2 > $OUT/td_$name.out 2> $OUT/td_$name.err
  echo "exit $?" | tee -a $S
  python -c "import json,sys; d=json.loads(open('$OUT/td_$name.out').readline()); print(round(d['value']), round(d['ms_per_step'],4), d['config']['parallelism'])" | tee -a $S
  grep -v "amdgpu.ids\|hostname of the client" $OUT/td_$name.err | tail -5 | tee -a $S
}
run seg1 200 FX_GRAPH_COLLECTIVES=0
run gc1 60 FX_GRAPH_COLLECTIVES=1
run seg2 60 FX_GRAPH_COLLECTIVES=0
run gc2 60 FX_GRAPH_COLLECTIVES=1
