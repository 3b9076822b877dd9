#!/usr/bin/env python3
"""FIRST-STEP GRADIENTS at the BASELINE shapes: one forward / backward from identical
weights on one seeded batch, every parameter gradient of
    native   the HIP path (unique-row table gradients scattered to dense, dense towers as they are)
    cpu32    the oracle on ATen's CPU kernels, fp32            (the reference's own arithmetic)
    gpu32    the oracle's identical code on ATen's GPU kernels (the reference with `gpu: 0`)
against the oracle evaluated in float64, as relative L2 error and as the largest element error in
units of ulp(max |g|) of the tensor.  Prints one JSON line per (case, dist, seed).

usage: grad_parity.py [--case c3_dcnv2] [--dist powerlaw] [--seeds 3,13,0,1] [--tag default]
Environment switches of the kernels under test (FX_DW_SPLITK, FX_GEMM_MULTI, FX_GEMM_PAIR, ...) are
read by the library at import: run one process per variant."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def errors(g, ref64):
    d = (g.double() - ref64).abs()
    nrm = float(ref64.norm())
    mx = float(ref64.abs().max())
    ulp = float(np.spacing(np.float32(mx))) if mx > 0 else 0.0
    return {"rel_l2": float(d.norm()) / nrm if nrm > 0 else float(d.norm()),
            "max_ulp": float(d.max()) / ulp if ulp > 0 else float(d.max())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="c3_dcnv2")
    ap.add_argument("--dist", default="powerlaw")
    ap.add_argument("--seeds", default="3,13,0,1")
    ap.add_argument("--tag", default="default")
    ap.add_argument("--batch", type=int, default=4096)
    args = ap.parse_args()
    import baseline_shapes as BS
    from fuxictr_amd import zoo
    from oracle import ctr_oracle as O
    for seed in [int(x) for x in args.seeds.split(",")]:
        model, features, cfg, spec, cards = BS.build(args.case, zoo, 0, "/tmp/fx_grad_parity", seed=seed)
        state0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        teacher = BS.Teacher(features)
        rng = np.random.default_rng({"powerlaw": 11, "uniform": 12}[args.dist])
        batch = BS.tb(BS.make_batches(args.case, spec, cards, rng, args.batch, 1, args.dist, teacher)[0])
        tr = O.OracleTrainer(cfg, state0, features, lr=1e-3, max_norm=10.0)
        g64 = tr.gradients(batch, batch["label"], double=True)
        g32 = tr.gradients(batch, batch["label"])
        trg = O.OracleTrainer(cfg, state0, features, lr=1e-3, max_norm=10.0, device="cuda:0")
        ggpu = trg.gradients(batch, batch["label"])
        gnat = BS.native_gradients(model, batch)
        rows = {}
        for k, ref in g64.items():
            if k not in gnat:
                raise SystemExit("native path has no gradient for %s" % k)
            ref = ref.double()
            rows[k] = {"native": errors(gnat[k].reshape(ref.shape), ref),
                       "cpu32": errors(g32[k], ref), "gpu32": errors(ggpu[k], ref),
                       "norm": float(ref.norm()), "numel": ref.numel()}
        # one summary row over all table gradients (26 x 2 tensors would drown the dense ones)
        print(json.dumps({"case": args.case, "dist": args.dist, "seed": seed, "tag": args.tag,
                          "tensors": rows}), flush=True)
        del model, tr, trg
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
