"""the c3 DCNv2 / power-law parity tail, taken apart at the GRADIENT level.

One forward / backward from identical weights on one seeded batch at the c3 shapes (3 cross layers
624 x 624 + MLP 4 x 1024, B = 4096, vocab x 0.01); every parameter gradient of the native path, of the
oracle on ATen's CPU kernels (fp32) and of the oracle on ATen's GPU kernels against the oracle evaluated
in float64 (scripts/grad_parity.py) — on the two seeds whose 10-step trajectories diverged most in round
3's sweep (3, 13) and on two well-behaved ones.  Asserted, per tensor: the native gradient is no further
from the float64 gradient than 2 x the worse of the reference's own two fp32 back ends (relative L2;
with a floor of 2e-7, one fp32 ulp-ish, for tensors both yardsticks happen to hit exactly).  The same
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
FLOOR = 2e-7


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
    worst = []
    for r in rows:
        for name, t in r["tensors"].items():
            yard = max(t["cpu32"]["rel_l2"], t["gpu32"]["rel_l2"], FLOOR)
            ratio = t["native"]["rel_l2"] / yard
            worst.append((ratio, r["seed"], r["tag"], name, t["native"]["rel_l2"], yard))
            assert ratio <= 2.0, ("native gradient further from fp64 than 2 x the reference's own back ends",
                                  r["seed"], r["tag"], name, t)
    worst.sort(reverse=True)
    print("[grad parity] worst ratios:", worst[:5])


def test_first_step_gradients_c3_powerlaw_seeds():
    _check(_run("3,13,0,1", "default"))


@pytest.mark.parametrize("tag,env", [("dw_no_split_k", {"FX_DW_SPLITK": "1"}),
                                     ("no_pair_no_multi", {"FX_GEMM_MULTI": "0", "FX_GEMM_PAIR": "0"})])
def test_first_step_gradients_c3_ab_of_the_summation_orders(tag, env):
    _check(_run("3,13", tag, env=env))


def test_first_step_gradients_c2_and_uniform():
    _check(_run("3", "default", case="c2_deepfm"))
    _check(_run("3", "default", dist="uniform"))
