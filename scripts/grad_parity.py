#!/usr/bin/env python3
"""FIRST-STEP GRADIENTS at the BASELINE shapes: one forward / backward from identical
weights on one seeded batch, every parameter gradient of
    native   the HIP path (unique-row table gradients scattered to dense, dense towers as they are)
    cpu32    the oracle on ATen's CPU kernels, fp32            (the reference's own arithmetic)
    gpu32    the oracle's identical code on ATen's GPU kernels (the reference with `gpu: 0`)
against the oracle evaluated in float64, as relative L2 error and as the largest element error in
units of ulp(max |g|) of the tensor.  Two comparisons per tensor: against the plain float64 gradient, and
against the float64 gradient evaluated on the evaluation's OWN ReLU decisions (`same_masks`): a hidden
unit whose pre-activation lies within rounding distance of zero flips its ReLU between any two fp32
evaluations, ONE such flip in the top hidden layer moves every gradient below it by ~1e-4 relative, and
how many there are is a lottery over ~17 M pre-activations (`relu_flips_vs_fp64_per_layer`) — the second
comparison takes that lottery out.  Prints one JSON line per (case, dist, seed).

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


class ReluTap(object):
    """Stand-in for the `F` the oracle calls: F.relu either records the mask (h > 0) of every call, in call
    order, or IMPOSES given masks (relu(h) := h * mask — the same value wherever the mask is right, and the
    mask's derivative), so that two evaluations can be compared on the SAME ReLU decisions."""

    def __init__(self, masks=None):
        self.masks, self.use, self.i = (masks or []), masks is not None, 0

    def __getattr__(self, name):
        return getattr(torch.nn.functional, name)

    def relu(self, h):
        if self.use:
            m = self.masks[self.i].to(device=h.device, dtype=h.dtype)
            self.i += 1
            return h * m
        self.masks.append((h > 0).cpu())
        return torch.nn.functional.relu(h)


def oracle_grads(O, tr, batch, double, masks=None):
    tap = ReluTap(masks)
    keep, O.F = O.F, tap
    try:
        g = tr.gradients(batch, batch["label"], double=double)
    finally:
        O.F = keep
    return g, tap.masks


def native_masks(model, case, batch):
    """The ReLU decisions of the NATIVE forward: the tower layer by layer on the native GEMM (the
    k-ordered fp32 MFMA chain the fused tower node runs; the ReLU epilogue does not enter the sums)."""
    from fuxictr_amd import ops
    tower = model.mlp if case == "c2_deepfm" else model.parallel_dnn
    model.eval()
    with torch.no_grad():
        X = model.get_inputs(batch)
        h = model.embedding_layer(X, flatten_emb=True).contiguous()
        masks = []
        lins = [m for m in tower.mlp if hasattr(m, "weight") and m.weight.dim() == 2]
        n_hidden = len(lins) - (1 if case == "c2_deepfm" else 0)
        for lin in lins[:n_hidden]:
            z = torch.empty(h.shape[0], lin.weight.shape[0], dtype=torch.float32, device=h.device)
            ops.gemm(h, lin.weight, z, transb=True, bias=lin.bias)
            masks.append((z > 0).cpu())
            h = torch.relu(z)
    model.train()
    return masks


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
        trg = O.OracleTrainer(cfg, state0, features, lr=1e-3, max_norm=10.0, device="cuda:0")
        g64, m64 = oracle_grads(O, tr, batch, True)
        g32, m32 = oracle_grads(O, tr, batch, False)
        ggpu, mgpu = oracle_grads(O, trg, batch, False)
        mnat = native_masks(model, args.case, batch) if args.case in ("c2_deepfm", "c3_dcnv2") else None
        gnat = BS.native_gradients(model, batch)
        # the float64 gradient on each evaluation's OWN ReLU decisions
        own = {"cpu32": oracle_grads(O, tr, batch, True, masks=m32)[0],
               "gpu32": oracle_grads(O, tr, batch, True, masks=mgpu)[0]}
        if mnat is not None and len(mnat) == len(m64):
            own["native"] = oracle_grads(O, tr, batch, True, masks=mnat)[0]
        flips = {"cpu32": [int((a != b).sum()) for a, b in zip(m32, m64)],
                 "gpu32": [int((a != b).sum()) for a, b in zip(mgpu, m64)]}
        if "native" in own:
            flips["native"] = [int((a != b).sum()) for a, b in zip(mnat, m64)]
            flips["native_vs_gpu32"] = [int((a != b).sum()) for a, b in zip(mnat, mgpu)]
        rows = {}
        for k, ref in g64.items():
            if k not in gnat:
                raise SystemExit("native path has no gradient for %s" % k)
            ref = ref.double()
            got = {"native": gnat[k].reshape(ref.shape), "cpu32": g32[k], "gpu32": ggpu[k]}
            rows[k] = {v: errors(got[v], ref) for v in got}
            rows[k].update(norm=float(ref.norm()), numel=ref.numel())
            # ... and against the float64 gradient evaluated on the SAME ReLU decisions
            rows[k]["same_masks"] = {v: errors(got[v], own[v][k].double()) for v in own}
        print(json.dumps({"case": args.case, "dist": args.dist, "seed": seed, "tag": args.tag,
                          "relu_flips_vs_fp64_per_layer": flips, "tensors": rows}), flush=True)
        del model, tr, trg
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
