#!/bin/bash
# kernel timeline of ONE steady-state step of bench.py (args passed through)
TAG=$1; shift
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$TAG -- \
    python $REPO/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-kernel-timing "$@" > /dev/null 2> $OUT/prof_$TAG.err)
TR=$(find /tmp/prof_$TAG -name '*kernel_trace.csv' | head -1)
python $REPO/scripts/step_timeline.py $TR 3 > $OUT/timeline_$TAG.txt
tail -1 $OUT/timeline_$TAG.txt
