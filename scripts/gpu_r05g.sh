#!/bin/bash
# grad parity (c3 seeds 1 and uniform 3) with the split-bf16 GEMM on / off + production GEMM bench with B prefetch
mkdir -p gpurun_out; export TMPDIR=/tmp
for x in 1 0; do
  FX_GEMM_BF16X6=$x timeout 600 python scripts/grad_parity.py --case c3_dcnv2 --seeds 1 --tag x6_$x > gpurun_out/r05_grad_parity_x6_$x.jsonl 2> gpurun_out/gp_$x.err
  FX_GEMM_BF16X6=$x timeout 600 python scripts/grad_parity.py --case c3_dcnv2 --dist uniform --seeds 3 --tag x6_$x >> gpurun_out/r05_grad_parity_x6_$x.jsonl 2>> gpurun_out/gp_$x.err
done
python - <<'PY'
import json
for x in (1, 0):
    for line in open("gpurun_out/r05_grad_parity_x6_%d.jsonl" % x):
        if not line.startswith("{"): continue
        r = json.loads(line)
        print("== x6=%d %s %s seed %d flips %s" % (x, r["case"], r["dist"], r["seed"], r["relu_flips_vs_fp64_per_layer"]))
        for name, t in r["tensors"].items():
            sm = t["same_masks"]
            print("  %-52s norm %.3e  same-mask rel_l2: native %.2e cpu32 %.2e gpu32 %.2e | plain native %.2e cpu32 %.2e" % (
                name[-52:], t["norm"], sm["native"]["rel_l2"], sm["cpu32"]["rel_l2"], sm["gpu32"]["rel_l2"], t["native"]["rel_l2"], t["cpu32"]["rel_l2"]))
PY
bash scripts/gpu_r05f.sh bdist
