"""The k-loop body selection of the pipelined GEMM (fx_gemm.hip, fx_gemm_pipe_tile) must not change a
single bit: tiles on the M / N edge on the unmasked bodies (FX_GEMM_EDGE_PLAIN=1, the round-4 default)
against the masked bodies wherever an edge is near (=0, rounds 1-3).  The switch is read once per
process: one subprocess per setting, the outputs compared bit for bit — on the shapes where edges matter (the 624-wide record:
first tower layer, CrossNetV2 layer; ragged M; K with and without a tail; K slabs)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys
import numpy as np
import torch
sys.path.insert(0, %r)
from fuxictr_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(7)
def rnd(*s):
    return torch.randn(*s, generator=g).to(dev)
out = {}
# (M, N_out, K_in): forward y = x W^T + b (K tail when K_in %% 32), then dW + dX pair
for tag, (M, N, K) in {"first": (4096, 1024, 624), "cross": (4096, 624, 624), "ragged": (1000, 520, 136),
                       "small": (332, 260, 72), "tower": (2048, 512, 1024), "one_tile": (64, 64, 64),
                       "two_tiles": (128, 64, 64)}.items():
    x, W, b, dz = rnd(M, K), rnd(N, K), rnd(N), rnd(M, N)
    y = torch.empty(M, N, device=dev)
    ops.gemm(x, W, y, transb=True, bias=b, act=1)
    out[tag + "/y"] = y.cpu().numpy()
    for sk in (1, 8):
        dW, dx, rs = torch.empty(N, K, device=dev), torch.empty(M, K, device=dev), torch.empty(N, device=dev)
        ws = torch.empty(ops.gemm_workspace_floats(N, K, sk), device=dev)
        ops.gemm_dw_dx(dz, x, W, dW, dx, split_k=sk, workspace=ws, rowsum=rs)
        out["%%s/dW%%d" %% (tag, sk)] = dW.cpu().numpy()
        out["%%s/dx%%d" %% (tag, sk)] = dx.cpu().numpy()
        out["%%s/db%%d" %% (tag, sk)] = rs.cpu().numpy()
        dW2 = torch.empty(N, K, device=dev)
        ops.gemm(dz, x, dW2, transa=True, split_k=sk, workspace=ws)
        out["%%s/dWsingle%%d" %% (tag, sk)] = dW2.cpu().numpy()
    # the oracle of last resort: float64 on the host
    ref = torch.relu(x.double().cpu() @ W.double().cpu().t() + b.double().cpu()).numpy()
    assert np.abs(out[tag + "/y"] - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max()), tag
    refdx = (dz.double().cpu() @ W.double().cpu()).numpy()
    assert np.abs(out[tag + "/dx1"] - refdx).max() <= 2e-4 * max(1.0, np.abs(refdx).max()), tag
    refdw = (dz.double().cpu().t() @ x.double().cpu()).numpy()
    assert np.abs(out[tag + "/dW8"] - refdw).max() <= 2e-4 * max(1.0, np.abs(refdw).max()), tag
np.savez(sys.argv[1], **out)
"""


def _run(mode, tmp_path, var="FX_GEMM_EDGE_PLAIN"):
    out = str(tmp_path / ("%s_%s.npz" % (var, mode)))
    env = dict(os.environ)
    env["FX_GEMM_BF16X6"] = "0"       # these switches select among the fp32-MFMA kernels' forms
    env[var] = mode
    p = subprocess.run([sys.executable, "-c", SCRIPT % ROOT, out], env=env, capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    return np.load(out)


def test_body_selection_is_bit_identical(tmp_path):
    base = _run("0", tmp_path)
    for mode in ("1",):
        z = _run(mode, tmp_path)
        assert sorted(z.files) == sorted(base.files)
        for k in base.files:
            assert np.array_equal(z[k].view(np.uint32), base[k].view(np.uint32)), (mode, k)


def test_vector_slab_reduce_is_bit_identical(tmp_path):
    """k_splitk_reduce_v4 (16-byte loads, all slabs of a vector in flight) adds the slabs in the order of
    the 4-byte kernel it replaces (FX_SPLITK_V4=0): weight gradients and bias gradients bit for bit."""
    base = _run("0", tmp_path, var="FX_SPLITK_V4")
    z = _run("1", tmp_path, var="FX_SPLITK_V4")
    for k in base.files:
        assert np.array_equal(z[k].view(np.uint32), base[k].view(np.uint32)), k
