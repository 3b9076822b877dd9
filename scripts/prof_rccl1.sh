#!/bin/bash
# rocprofv3 kernel stats of the row-sharded step on ONE rank through the RCCL backend
# (FX_SHARD_WORLD1=1): every kernel of the sharded pipeline including RCCL's own.
REPO=$PWD
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
rm -rf $OUT/prof_rccl1
(cd /tmp && FX_SHARD_WORLD1=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_rccl1 -- \
    python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > $OUT/prof_rccl1.json 2> $OUT/prof_rccl1.err)
echo "exit $?"
tail -1 $OUT/prof_rccl1.json | cut -c1-200
STATS=$(find $OUT/prof_rccl1 -name '*kernel_stats.csv' | head -1)
cp $STATS $OUT/kernel_stats_rccl1.csv
python $REPO/scripts/kstats.py $STATS 25 40
TRACE=$(find $OUT/prof_rccl1 -name '*kernel_trace.csv' | head -1)
python $REPO/scripts/step_timeline.py $TRACE > $OUT/timeline_rccl1.txt 2>&1; tail -70 $OUT/timeline_rccl1.txt
find $OUT/prof_rccl1 -name '*kernel_trace.csv' -size +20M -delete
