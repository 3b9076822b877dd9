#!/bin/bash
# rocprofv3 kernel stats of bench.py (args passed through) -> gpurun_out/kstats_<tag>.csv + summary
TAG=$1; shift
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rm -rf $OUT/prof_$TAG
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- \
    python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing "$@" > $OUT/prof_bench_$TAG.json 2> $OUT/prof_$TAG.err)
STATS=$(find $OUT/prof_$TAG -name '*kernel_stats.csv' | head -1)
cp $STATS $OUT/kstats_$TAG.csv
find $OUT/prof_$TAG -name '*kernel_trace.csv' -delete
python $REPO/scripts/kstats.py $OUT/kstats_$TAG.csv 25 ${TOPN:-12}
