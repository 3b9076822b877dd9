"""the c3 DCNv2 / power-law parity tail, taken apart at the GRADIENT level.

One forward / backward from identical weights on one seeded batch at the c3 shapes (3 cross layers
624 x 624 + MLP 4 x 1024, B = 4096, vocab x 0.01); every parameter gradient of the native path, of the
oracle on ATen's CPU kernels (fp32) and of the oracle on ATen's GPU kernels against the oracle evaluated
in float64 (scripts/grad_parity.py) — on the two seeds whose 10-step trajectories diverged most in round
3's sweep (3, 13) and on two well-behaved ones.  What the first run of this test showed (profiles/
r04_grad_parity_*.txt): every fp32 evaluation's tower gradients sit 1e-5 .. 1e-4 from the float64 ones,
and that distance is ReLU decisions — a pre-activation within rounding distance of zero flips, one flip in
the top hidden layer moves every gradient below it by ~1e-4 — not summation order.  So the assertion is
made on the SAME ReLU decisions: against the float64 gradient evaluated with the evaluation's own masks,
the native gradient is no further away than 2 x the worse of the reference's own two fp32 back ends, per
tensor (relative L2, floor 1e-6); the flip counts per layer and the plain ratios are printed.  The same
for the A/B forms of the two places where the native summation order differs structurally: the weight
gradient without K slabs (FX_DW_SPLITK=1) and the pair / multi grids off (FX_GEMM_MULTI=0
FX_GEMM_PAIR=0).  The per-tensor table goes to FX_GRAD_PARITY_REPORT (committed under profiles/)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLOOR = 1e-6


def _run(seeds, tag, env=None, case="c3_dcnv2", dist="powerlaw"):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "grad_parity.py"), "--case", case,
                        "--dist", dist, "--seeds", seeds, "--tag", tag], capture_output=True, text=True,
                       env=e, timeout=900)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    rows = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    out = os.environ.get("FX_GRAD_PARITY_REPORT")
    if out:
        with open(out, "a") as f:
            for r in rows:
                f.write(json.dumps(r) + "\n")
    return rows


def _check(rows):
    """Per tensor, on the evaluation's own ReLU decisions: native <= 2 x the worse yardstick.  The plain
    comparison (ReLU flips included) is reported, not asserted: one flip in the top hidden layer is
    ~1e-4 in every tensor below it, and whether an evaluation draws 0, 1 or 3 of them is chance."""
    worst, plain = [], []
    for r in rows:
        print("[grad parity] %s %s seed %d [%s]: ReLU flips vs fp64 per layer %s"
              % (r["case"], r["dist"], r["seed"], r["tag"], r["relu_flips_vs_fp64_per_layer"]))
        for name, t in r["tensors"].items():
            sm = t["same_masks"]
            assert "native" in sm, "no native ReLU masks recorded"
            yard = max(sm["cpu32"]["rel_l2"], sm["gpu32"]["rel_l2"], FLOOR)
            ratio = sm["native"]["rel_l2"] / yard
            worst.append((round(ratio, 2), r["seed"], name[-40:], sm["native"]["rel_l2"], yard))
            plain.append((round(t["native"]["rel_l2"] /
                                max(t["cpu32"]["rel_l2"], t["gpu32"]["rel_l2"], FLOOR), 2), r["seed"],
                          name[-40:]))
            assert ratio <= 2.0, ("native gradient further from the fp64 gradient (same ReLU decisions) than "
                                  "2 x the reference's own back ends", r["seed"], r["tag"], name, t)
    worst.sort(reverse=True)
    plain.sort(reverse=True)
    print("[grad parity] worst ratios on the same ReLU decisions:", worst[:4])
    print("[grad parity] worst ratios, flips included (not asserted):", plain[:4])


def test_first_step_gradients_c3_powerlaw_seeds():
    _check(_run("3,13,0,1", "default"))


@pytest.mark.parametrize("tag,env", [("dw_no_split_k", {"FX_DW_SPLITK": "1"}),
                                     ("no_pair_no_multi", {"FX_GEMM_MULTI": "0", "FX_GEMM_PAIR": "0"})])
def test_first_step_gradients_c3_ab_of_the_summation_orders(tag, env):
    _check(_run("3,13", tag, env=env))


def test_first_step_gradients_c2_and_uniform():
    _check(_run("3", "default", case="c2_deepfm"))
    _check(_run("3", "default", dist="uniform"))
