#!/usr/bin/env python3
"""Timing lab for the CIN kernels (fx_cin_fwd / fx_cin_bwd) on the BASELINE xDeepFM shapes: layer 1
(Mi = 39) and layers 2/3 (Mi = 16) at several batch sizes; back-to-back launches timed with events on
the launch stream.  usage: cin_lab.py [B ...]   env FX_CIN_MFMA=0 times the VALU kernels."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fuxictr_amd import ops  # noqa: E402

DEV = torch.device("cuda:0")
REPS = int(os.environ.get("CIN_LAB_REPS", "30"))


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / REPS


def main():
    Bs = [int(x) for x in sys.argv[1:]] or [4096, 8192, 16384]
    F0, D, O = 39, 16, 16
    print("%-6s %-4s %10s %10s %10s   (us per launch; MFMA floor at 157.3 TF)" % ("B", "Mi", "fwd", "bwd", "floor"))
    for B in Bs:
        for Mi in (39, 16):
            g = torch.Generator().manual_seed(1)
            x0 = torch.randn(B, F0, D, generator=g).to(DEV)
            xi = torch.randn(B, Mi, D, generator=g).to(DEV)
            W = (torch.randn(O, F0 * Mi, generator=g) * 0.05).to(DEV)
            bias = torch.zeros(O, device=DEV)
            xn = torch.empty(B, O, D, device=DEV)
            pool = torch.empty(B, O, device=DEV)
            gxn = torch.randn(B, O, D, generator=g).to(DEV)
            gpool = torch.randn(B, O, generator=g).to(DEV)
            dx0 = torch.zeros(B, F0, D, device=DEV)
            dxi = torch.empty(B, Mi, D, device=DEV)
            partial = torch.empty(ops.cin_workgroups(), O * F0 * Mi + O, device=DEV)
            n = ops.cin_wimg_floats(F0, Mi, D, O)
            img = torch.empty(n, device=DEV) if n else None
            if n:
                ops.cin_pack_w([(W, F0, Mi, img)], D)
            tf = timed(lambda: ops.cin_fwd(x0, xi, W, bias, xn, pool, img))
            tb = timed(lambda: ops.cin_bwd(x0, xi, W, gxn, gpool, dx0, Mi == 16, dxi, partial, img))
            floor = 2.0 * B * O * F0 * Mi * D / 157.3e12 * 1e6
            print("%-6d %-4d %10.1f %10.1f %10.1f" % (B, Mi, tf, tb, floor))


if __name__ == "__main__":
    main()
