#!/bin/bash
# one-step kernel timelines (rocprofv3 kernel trace) of the given models at HEAD.  usage: gpu_timelines.sh TAG MODEL...
TAG=$1; shift
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for M in "$@"; do
  rm -rf /tmp/prof_${TAG}_$M
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_$M -- \
      python $REPO/bench.py --model $M --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-dcnv2 --no-din --no-parity --no-uniform > /dev/null 2> $OUT/prof_${TAG}_$M.err)
  TR=$(find /tmp/prof_${TAG}_$M -name '*kernel_trace.csv' | head -1)
  python scripts/step_timeline.py $TR 3 > $OUT/timeline_${M}_$TAG.txt; echo "$M $(tail -1 $OUT/timeline_${M}_$TAG.txt)"
  ST=$(find /tmp/prof_${TAG}_$M -name '*kernel_stats.csv' | head -1); [ -n "$ST" ] && cp $ST $OUT/kernel_stats_${M}_$TAG.csv
done
