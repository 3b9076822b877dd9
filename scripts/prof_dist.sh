#!/bin/bash
# rocprofv3 kernel stats of the row-sharded step with 2 ranks on ONE GPU (collectives staged through
# gloo): gives the GPU-side kernel time of the sharded pipeline (everything except RCCL itself).
REPO=$PWD
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
rm -rf $OUT/prof_dist
(cd /tmp && FX_BENCH_BACKEND=gloo timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_dist -- \
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
    $REPO/bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > $OUT/prof_dist.json 2> $OUT/prof_dist.err)
echo "exit $?"
tail -1 $OUT/prof_dist.json | cut -c1-300
for f in $(find $OUT/prof_dist -name '*kernel_stats.csv'); do echo "== $f"; head -45 $f; done
find $OUT/prof_dist -name '*kernel_trace.csv' -size +20M -delete
