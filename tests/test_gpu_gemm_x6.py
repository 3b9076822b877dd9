"""The split-bf16 GEMM (fx_gemm_x6.hip: every fp32 operand split into three exact bf16 planes inside the kernel,
six v_mfma_f32_32x32x16_bf16 products per fp32 product, fp32 accumulate) behind fx_gemm_f32 / fx_gemm_f32_batch.

It computes the products of mlp_block.py:96 and cross_net.py:126-129 (and their autograd, rank_model.py:320) the
reference gets from aten::addmm in fp32; the claim is "fp32 accuracy", so every check is against float64 with
a bound TIGHTER than the one the fp32-MFMA kernels are held to (tests/test_gpu_kernels.py: 2e-6 x bound):
all four operand layouts, the M / N edges and K tails of the 624-wide record, K slabs with the fused bias
gradient, every epilogue operand, the multi-problem grid.  A subprocess pair (FX_GEMM_BF16X6=0 / 1) shows that
the switch selects different kernels (different bits on the large shapes) whose results agree to fp32 rounding.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = torch.device("cuda:0") if torch.cuda.is_available() else None


def _ops():
    from fuxictr_amd import ops
    return ops


def _dev(t):
    return t.to(DEV)


# (M, N, K): >= 96 tiles of 128 x 128 (x slabs) so that fx_gemm_f32 takes the split-bf16 kernels
X6_SHAPES = [(4096, 1024, 1024), (4096, 1024, 624), (4096, 624, 624), (4096, 624, 1024), (2000, 1160, 200),
             (4096, 628, 68), (4092, 628, 68), (1536, 1536, 100)]


@pytest.mark.parametrize("M,N,K", X6_SHAPES)
@pytest.mark.parametrize("ta,tb", [(False, True), (False, False), (True, False), (True, True)])
def test_x6_all_layouts_against_float64(M, N, K, ta, tb):
    ops = _ops()
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    A = torch.randn((K, M) if ta else (M, K), generator=g)
    Bm = torch.randn((N, K) if tb else (K, N), generator=g) * 0.1
    C = torch.full((M, N), float("nan"), device=DEV)
    ops.gemm(_dev(A), _dev(Bm), C, transa=ta, transb=tb)
    a = A.t() if ta else A
    b = Bm.t() if tb else Bm
    ref = a.double() @ b.double()
    bound = (a.abs().double() @ b.abs().double()).max().item()
    err = (C.cpu().double() - ref).abs().max().item()
    assert err <= 1e-6 * bound, (err, bound)
    rel = ((C.cpu().double() - ref).norm() / ref.norm()).item()
    assert rel <= 1.5e-6, rel


@pytest.mark.parametrize("M,N,K,sk", [(1024, 1024, 4096, 4), (1024, 624, 4096, 8), (624, 624, 4096, 8),
                                      (1024, 1024, 4000, 5), (640, 1024, 4100, 3),
                                      # no K split: the row sums leave the kernel directly; slabs of one and two
                                      # k tiles, the last one partial (the pipeline's fill is then most of the loop)
                                      (1536, 1536, 512, 1), (2048, 1280, 200, 5), (1536, 1536, 72, 2)])
def test_x6_weight_gradient_slabs_and_bias_gradient(M, N, K, sk):
    """dW[M, N] = dz^T x with K slabs and the fused row sums of dz^T (the bias gradient), ragged K included."""
    ops = _ops()
    g = torch.Generator().manual_seed(M + K + sk)
    dz = torch.randn(K, M, generator=g)
    x = torch.randn(K, N, generator=g)
    dW = torch.full((M, N), float("nan"), device=DEV)
    db = torch.full((M,), float("nan"), device=DEV)
    ws = torch.empty(ops.gemm_workspace_floats(M, N, sk), device=DEV)
    ops.gemm(_dev(dz), _dev(x), dW, transa=True, split_k=sk, workspace=ws, rowsum=db)
    ref = dz.double().t() @ x.double()
    bound = (dz.abs().double().t() @ x.abs().double()).max().item()
    assert (dW.cpu().double() - ref).abs().max().item() <= 1e-6 * bound
    refb = dz.double().sum(0)
    assert (db.cpu().double() - refb).abs().max().item() <= 1e-6 * dz.abs().double().sum(0).max().item()


def test_x6_epilogues():
    """bias, pre-activation output, ReLU, Hadamard operand, mask, residual — into a strided output."""
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    M, N, K = 4096, 624, 624
    x, W = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.05
    bias, mul, add, msk = (torch.randn(N, generator=g), torch.randn(M, N, generator=g),
                           torch.randn(M, N, generator=g), torch.randn(M, N, generator=g))
    z_ref = x.double() @ W.double().t() + bias.double()
    tol = 1.5e-6 * (x.abs().double() @ W.abs().double().t()).max().item()
    wide = torch.full((M, N + 1024), float("nan"), device=DEV)
    C, Z = wide[:, :N], torch.empty(M, N, device=DEV)
    ops.gemm(_dev(x), _dev(W), C, transb=True, bias=_dev(bias), act=1)
    assert (C.cpu().double() - z_ref.clamp(min=0)).abs().max().item() <= tol
    ops.gemm(_dev(x), _dev(W), C, transb=True, bias=_dev(bias), zout=Z, mul=_dev(mul), add=_dev(add))
    assert (Z.cpu().double() - z_ref).abs().max().item() <= tol
    assert (C.cpu().double() - (z_ref * mul.double() + add.double())).abs().max().item() <= 4 * tol
    ops.gemm(_dev(x), _dev(W), C, transb=True, mask=_dev(msk))
    ref = torch.where(msk > 0, x.double() @ W.double().t(), torch.zeros((), dtype=torch.float64))
    assert (C.cpu().double() - ref).abs().max().item() <= tol
    assert torch.isnan(wide[:, N:]).all()                    # nothing past the output's columns was touched


def test_x6_exact_cases():
    """Products that are exact in fp32 must come out exact: the three planes of an operand add up to it bit
    for bit (A = I picks single elements of an asymmetric B: also catches a transposed fragment mapping), and
    small integers stay integers."""
    ops = _ops()
    n = 2048
    Bm = (torch.arange(n * n, dtype=torch.float32).view(n, n) * 1.000123 - 999.5)
    C = torch.empty(n, n, device=DEV)
    ops.gemm(_dev(torch.eye(n)), _dev(Bm), C)
    assert torch.equal(C.cpu(), Bm)
    g = torch.Generator().manual_seed(1)
    A = torch.randint(-8, 9, (2048, 256), generator=g).float()
    W = torch.randint(-8, 9, (1536, 256), generator=g).float()
    C = torch.empty(2048, 1536, device=DEV)
    ops.gemm(_dev(A), _dev(W), C, transb=True)
    assert torch.equal(C.cpu(), A @ W.t())


def test_x6_multi_problem_grid_equals_single_launches():
    """fx_gemm_f32_batch: DCNv2's four backward products of one depth in ONE grid — the same tile function,
    k order and (for the un-split dX) the same bits as single launches; dW against float64."""
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    M = 4096
    probs, outs, singles, keep = [], [], [], []
    for N, K in ((624, 624), (1024, 624)):
        dz, x = _dev(torch.randn(M, N, generator=g)), _dev(torch.randn(M, K, generator=g))
        W = _dev(torch.randn(N, K, generator=g) * 0.1)
        dW, dx = torch.full((N, K), float("nan"), device=DEV), torch.full((M, K), float("nan"), device=DEV)
        db = torch.full((N,), float("nan"), device=DEV)
        ws = torch.empty(ops.gemm_workspace_floats(N, K, 8), device=DEV)
        probs.append(ops.gemm_problem(dz, x, dW, transa=True, transb=False, split_k=8, workspace=ws, rowsum=db))
        probs.append(ops.gemm_problem(dz, W, dx, transa=False, transb=False))
        dx1 = torch.empty(M, K, device=DEV)
        ops.gemm(dz, W, dx1, transa=False, transb=False)
        outs.append((dz, x, W, dW, dx, db))
        singles.append(dx1)
        keep.append(ws)
    ops.gemm_batch(probs)
    torch.cuda.synchronize()
    for (dz, x, W, dW, dx, db), dx1 in zip(outs, singles):
        assert torch.equal(dx, dx1)
        ref_w = dz.double().t() @ x.double()
        assert (dW.double() - ref_w).abs().max().item() <= 1e-6 * (dz.abs().double().t() @ x.abs().double()).max().item()
        assert (db.double() - dz.double().sum(0)).abs().max().item() <= 1e-6 * dz.abs().double().sum(0).max().item()


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
for tag, (M, N, K) in {"first": (4096, 1024, 624), "cross": (4096, 624, 624), "tower": (4096, 1024, 1024)}.items():
    x, W, b, dz = rnd(M, K), rnd(N, K) * 0.1, rnd(N), rnd(M, N)
    y = torch.empty(M, N, device=dev)
    ops.gemm(x, W, y, transb=True, bias=b, act=1)
    out[tag + "/y"] = y.cpu().numpy()
    dW, dx, rs = torch.empty(N, K, device=dev), torch.empty(M, K, device=dev), torch.empty(N, device=dev)
    ws = torch.empty(ops.gemm_workspace_floats(N, K, 8), device=dev)
    ops.gemm_dw_dx(dz, x, W, dW, dx, split_k=8, workspace=ws, rowsum=rs)
    out[tag + "/dW"], out[tag + "/dx"], out[tag + "/db"] = dW.cpu().numpy(), dx.cpu().numpy(), rs.cpu().numpy()
    out[tag + "/ref_y"] = torch.relu(x.double() @ W.double().t() + b.double()).cpu().numpy()
    out[tag + "/ref_dW"] = (dz.double().t() @ x.double()).cpu().numpy()
    out[tag + "/ref_dx"] = (dz.double() @ W.double()).cpu().numpy()
np.savez(sys.argv[1], **out)
"""


def test_x6_switch_selects_other_kernels_with_the_same_results(tmp_path):
    res = {}
    for mode in ("0", "1"):
        out = str(tmp_path / ("x6_%s.npz" % mode))
        env = dict(os.environ)
        env["FX_GEMM_BF16X6"] = mode
        p = subprocess.run([sys.executable, "-c", SCRIPT % ROOT, out], env=env, capture_output=True, text=True,
                           timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        res[mode] = np.load(out)
    for tag in ("first", "cross", "tower"):
        for k in ("y", "dW", "dx"):
            a, b, ref = res["0"]["%s/%s" % (tag, k)], res["1"]["%s/%s" % (tag, k)], res["1"]["%s/ref_%s" % (tag, k)]
            assert not np.array_equal(a, b), (tag, k, "FX_GEMM_BF16X6 made no difference: which kernel ran?")
            e0 = np.linalg.norm(a - ref) / np.linalg.norm(ref)
            e1 = np.linalg.norm(b - ref) / np.linalg.norm(ref)
            print("[x6 a/b] %s/%s relative L2 error vs float64: fp32-MFMA %.3e  split-bf16 %.3e" % (tag, k, e0, e1))
            assert e1 <= 1.2e-6 and e1 <= 1.25 * e0 + 1e-8, (tag, k, e0, e1)


# ---- adversarial operand ranges: every test above draws randn ---------------------------------
RANGE_SCRIPT = r"""
import sys
import numpy as np
import torch
sys.path.insert(0, %r)
from fuxictr_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(17)
out = {}
M, N, K = 4096, 1024, 1024
def run(tag, x, W, dz):
    # forward x W^T, and the weight-gradient / input-gradient pair of the same layer (K slabs + fused row sums)
    y = torch.empty(M, N, device=dev)
    ops.gemm(x.to(dev), W.to(dev), y, transb=True)
    dW, dx, rs = torch.empty(N, K, device=dev), torch.empty(M, K, device=dev), torch.empty(N, device=dev)
    ws = torch.empty(ops.gemm_workspace_floats(N, K, 4), device=dev)
    ops.gemm_dw_dx(dz.to(dev), x.to(dev), W.to(dev), dW, dx, split_k=4, workspace=ws, rowsum=rs)
    out[tag + "/y"], out[tag + "/dW"], out[tag + "/dx"] = y.cpu().numpy(), dW.cpu().numpy(), dx.cpu().numpy()
x, W, dz = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.05, torch.randn(M, N, generator=g)
np.savez(sys.argv[1] + ".in", x=x.numpy(), W=W.numpy(), dz=dz.numpy())
for tag, sx, sw, sz in (("unit", 1.0, 1.0, 1.0), ("grad_1e-8", 1.0, 1.0, 1e-8), ("act_1e-20", 1e-20, 1.0, 1e-12),
                        ("tiny_1e-30", 1e-30, 1.0, 1.0), ("tiny_1e-35", 1e-35, 1.0, 1.0),
                        ("huge_1e30", 1e30, 1e5, 1e-25), ("mixed", 1e18, 1e-18, 1e-10)):
    run(tag, x * sx, W * sw, dz * sz)
# one Inf activation and one NaN upstream gradient: what leaves the rows / columns they touch
xi = x.clone(); xi[7, 100] = float("inf")
dzi = dz.clone(); dzi[11, 5] = float("nan")
run("nonfinite", xi, W, dzi)
np.savez(sys.argv[1], **out)
"""


def test_x6_operand_ranges_against_float64_and_the_fp32_kernels(tmp_path):
    """Scales the training step really produces (mean-BCE gradients of 1e-8, activations and weights of order
    one) and scales it never does (1e-35: the third bf16 plane of an operand is a bf16 denormal; 1e30), both
    kernels in separate processes, both against float64.  Inf / NaN: confined to the rows and columns they
    touch; where the fp32 chain would give Inf the split product gives NaN (0 x Inf between planes) —
    documented in fx_gemm_x6.hip and INTEGRATION.md, asserted here."""
    res = {}
    for mode in ("0", "1"):
        out = str(tmp_path / ("rng_%s.npz" % mode))
        env = dict(os.environ)
        env["FX_GEMM_BF16X6"] = mode
        p = subprocess.run([sys.executable, "-c", RANGE_SCRIPT % ROOT, out], env=env, capture_output=True,
                           text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-3000:]
        res[mode] = np.load(out)
    base = np.load(str(tmp_path / "rng_1.npz.in.npz"))
    x, W, dz = (base[k].astype(np.float64) for k in ("x", "W", "dz"))
    scales = {"unit": (1.0, 1.0, 1.0), "grad_1e-8": (1.0, 1.0, 1e-8), "act_1e-20": (1e-20, 1.0, 1e-12),
              "tiny_1e-30": (1e-30, 1.0, 1.0), "tiny_1e-35": (1e-35, 1.0, 1.0), "huge_1e30": (1e30, 1e5, 1e-25),
              "mixed": (1e18, 1e-18, 1e-10)}
    # float32 images of the scaled operands are what both kernels were given
    for tag, (sx, sw, sz) in scales.items():
        xs = (base["x"] * np.float32(sx)).astype(np.float64)
        Ws = (base["W"] * np.float32(sw)).astype(np.float64)
        zs = (base["dz"] * np.float32(sz)).astype(np.float64)
        refs = {"y": xs @ Ws.T, "dW": zs.T @ xs, "dx": zs @ Ws}
        for k, ref in refs.items():
            e0 = np.linalg.norm(res["0"]["%s/%s" % (tag, k)] - ref) / np.linalg.norm(ref)
            e1 = np.linalg.norm(res["1"]["%s/%s" % (tag, k)] - ref) / np.linalg.norm(ref)
            print("[x6 range] %-11s %-2s relative L2 error vs float64: fp32-MFMA %.3e  split-bf16 %.3e" % (tag, k, e0, e1))
            assert np.isfinite(res["1"]["%s/%s" % (tag, k)]).all(), (tag, k)
            if tag == "tiny_1e-35":
                # the third plane (2^-16 of 1e-35) lies below the smallest normal bf16 (1.2e-38): if the matrix
                # core flushes it the product keeps 16 significand bits — a documented limit far outside any
                # training state; bounded here so that a change of behaviour is noticed
                assert e1 <= 2e-5, (tag, k, e1)
            else:
                assert e1 <= 1.2e-6 and e1 <= 1.25 * e0 + 1e-8, (tag, k, e0, e1)
    # non-finite operands
    for mode in ("0", "1"):
        y, dW, dx = (res[mode]["nonfinite/" + k] for k in ("y", "dW", "dx"))
        assert not np.isfinite(y[7]).any() or mode == "0"            # row 7 of y saw the Inf activation
        assert np.isfinite(np.delete(y, 7, axis=0)).all(), mode      # ... and only row 7
        assert not np.isfinite(dx[11]).any() and np.isfinite(np.delete(dx, 11, axis=0)).all(), mode
        # dW[n, k] = sum_m dz[m, n] x[m, k]: NaN in column n = 5 of dz -> row 5 of dW; Inf in x[7, 100] -> column 100
        bad = ~np.isfinite(dW)
        assert bad[5].all() and bad[:, 100].all(), mode
        bad[5] = False
        bad[:, 100] = False
        assert not bad.any(), mode
    y0, y1 = res["0"]["nonfinite/y"][7], res["1"]["nonfinite/y"][7]
    print("[x6 range] Inf activation: fp32-MFMA row has %d Inf / %d NaN, split-bf16 row has %d Inf / %d NaN"
          % (np.isinf(y0).sum(), np.isnan(y0).sum(), np.isinf(y1).sum(), np.isnan(y1).sum()))
    assert np.isnan(y1).all()                                        # the documented Inf -> NaN of the split
