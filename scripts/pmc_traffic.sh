#!/bin/bash
# Fabric-side traffic of the kernels inside the DeepFM step, per launch: rocprofv3 --pmc in SEPARATE
# passes (FETCH_SIZE, WRITE_SIZE), kernel-trace only (MI355X_MICROARCH.md, HBM section).  Eager launches
# so that every dispatch is attributed.  -> gpurun_out/pmc_traffic_$TAG.{txt,json}
TAG=${1:-r05}
REPO=$PWD; OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$C -- \
      python $REPO/bench.py --steps 4 --warmup 5 --no-graph --no-step-events --no-cpu-baseline --no-kernel-timing --no-dcnv2 --no-din --no-parity --no-uniform > /dev/null 2> $OUT/pmc_$C.err)
done
python scripts/pmc_traffic.py "$(find /tmp/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)" \
    "$(find /tmp/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)" 9 $OUT/pmc_traffic_$TAG.json | tee $OUT/pmc_traffic_$TAG.txt
