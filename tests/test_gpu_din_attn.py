"""DIN target attention with the attention MLP fused in (fx_din_attn.hip) on a real MI355X, against
the oracle's restatement of target_attention.py:66-92 + activations.py:40-51 in fp64 (forward,
running statistics, every gradient), against the unfused native kernels, for the shapes of
BASELINE configs[3] (B = 4096, L = 50, E = 16, H = 64) and the ragged / padded cases: E not a
multiple of 8 (scalar row loads), H < 32 and 32 < H < 64 (padded MFMA blocks), L = 1, B*L not a
multiple of the 32-position tile, strided K (a view of the gather record), no mask, no biases,
eval mode.  Tolerances are fp32 summation-order bounds, written at each check."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from fuxictr_amd import layers as nat, ops  # noqa: E402
from oracle import ctr_oracle as O  # noqa: E402

DEV = "cuda:0"
PFX = "att."


def _case(B, L, E, H, seed, mask_p=0.3):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(B, E, generator=g)
    K = torch.randn(B, L, E, generator=g) * 0.8
    mask = (torch.rand(B, L, generator=g) > mask_p)
    state = {
        PFX + "attention_layer.mlp.0.weight": torch.randn(H, 4 * E, generator=g) * 0.3,
        PFX + "attention_layer.mlp.0.bias": torch.randn(H, generator=g) * 0.2,
        PFX + "attention_layer.mlp.1.bn.running_mean": torch.randn(H, generator=g) * 0.1,
        PFX + "attention_layer.mlp.1.bn.running_var": torch.rand(H, generator=g) + 0.5,
        PFX + "attention_layer.mlp.1.bn.num_batches_tracked": torch.zeros((), dtype=torch.long),
        PFX + "attention_layer.mlp.1.alpha": torch.rand(H, generator=g) - 0.5,
        PFX + "attention_layer.mlp.2.weight": torch.randn(1, H, generator=g) * 0.4,
        PFX + "attention_layer.mlp.2.bias": torch.randn(1, generator=g) * 0.1,
    }
    dout = torch.randn(B, E, generator=g)
    return q, K, mask, state, dout


def _oracle(q, K, mask, state, dout, training):
    st = {k: (v.double().clone() if v.is_floating_point() else v.clone()) for k, v in state.items()}
    leaves = {}
    for k in list(st):
        if k.endswith(("weight", "bias", "alpha")):
            st[k] = st[k].requires_grad_(True)
            leaves[k] = st[k]
    qd, Kd = q.double().requires_grad_(True), K.double().requires_grad_(True)
    out = O.din_attention(st, PFX, qd, Kd, mask, training)
    out.backward(dout.double())
    grads = {k[len(PFX) + len("attention_layer."):]: v.grad for k, v in leaves.items()}
    return out.detach(), qd.grad, Kd.grad, grads, st


def _module(E, H, state, fused):
    old = os.environ.get("FX_DIN_FUSED")
    os.environ["FX_DIN_FUSED"] = "1" if fused else "0"
    try:
        torch.cuda.set_device(0)
        mod = nat.DIN_Attention(embedding_dim=E, attention_units=[H], hidden_activations="Dice")
        mod.load_state_dict({k[len(PFX):]: v for k, v in state.items()})
        assert (mod._fused_plan() is not None) == fused
    finally:
        if old is None:
            os.environ.pop("FX_DIN_FUSED", None)
        else:
            os.environ["FX_DIN_FUSED"] = old
    return mod


def _run(mod, q, K, mask, dout, training, record_view=False):
    mod.train(training)
    qg = q.to(DEV).requires_grad_(True)
    if record_view:                                   # K = slots 2 .. 2+L of a wider record
        B, L, E = K.shape
        rec = torch.zeros(B, L + 5, E)
        rec[:, 2:2 + L] = K
        recg = rec.to(DEV).requires_grad_(True)
        Kg = recg[:, 2:2 + L, :]
    else:
        recg = None
        Kg = K.to(DEV).requires_grad_(True)
    out = mod(qg, Kg, None if mask is None else mask.to(DEV))
    out.backward(dout.to(DEV))
    dK = recg.grad[:, 2:2 + K.shape[1]] if record_view else Kg.grad
    grads = {n: p.grad for n, p in mod.attention_layer.named_parameters()}
    return out.detach(), qg.grad, dK, grads


def _close(got, ref, tol, what):
    ref = ref.to(torch.float64)
    err = (got.detach().cpu().double().reshape(ref.shape) - ref).abs().max().item()
    bound = tol * max(1.0, ref.abs().max().item())
    assert err <= bound, "%s: max |diff| %.3e > %.3e" % (what, err, bound)


SHAPES = [(64, 50, 16, 64), (7, 3, 4, 16), (33, 50, 8, 36), (5, 1, 10, 64), (300, 20, 16, 32),
          (50, 32, 16, 32), (41, 97, 8, 64), (19, 33, 16, 64), (2, 64, 16, 64),
          (3, 32, 8, 7),
          (129, 7, 12, 64), (4096, 50, 16, 64)]


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("B,L,E,H", SHAPES)
def test_fused_din_attention_matches_the_oracle(B, L, E, H, training):
    q, K, mask, state, dout = _case(B, L, E, H, seed=B + L + E + H)
    ro, rdq, rdK, rg, rst = _oracle(q, K, mask, state, dout, training)
    mod = _module(E, H, state, fused=True)
    out, dq, dK, grads = _run(mod, q, K, mask, dout, training)
    _close(out, ro, 2e-5, "out")
    _close(dq, rdq, 1e-4, "dq")
    _close(dK, rdK, 1e-4, "dK")
    for name in ("mlp.0.weight", "mlp.0.bias", "mlp.1.alpha", "mlp.2.weight", "mlp.2.bias"):
        _close(grads[name], rg[name], 2e-4, "d " + name)
    dice = mod.attention_layer.mlp[1]
    _close(dice.bn.running_mean, rst[PFX + "attention_layer.mlp.1.bn.running_mean"], 1e-6, "running_mean")
    _close(dice.bn.running_var, rst[PFX + "attention_layer.mlp.1.bn.running_var"], 1e-5, "running_var")
    assert int(dice.bn.num_batches_tracked) == (1 if training else 0)


@pytest.mark.parametrize("B,L,E,H", [(257, 50, 16, 64), (40, 6, 8, 16)])
def test_fused_and_unfused_native_paths_agree(B, L, E, H):
    q, K, mask, state, dout = _case(B, L, E, H, seed=11)
    a = _run(_module(E, H, state, fused=True), q, K, mask, dout, True)
    b = _run(_module(E, H, state, fused=False), q, K, mask, dout, True)
    _close(a[0], b[0].cpu(), 1e-5, "out")
    _close(a[1], b[1].cpu(), 5e-5, "dq")
    _close(a[2], b[2].cpu(), 5e-5, "dK")
    for name in a[3]:
        _close(a[3][name], b[3][name].cpu(), 1e-4, "d " + name)


@pytest.mark.parametrize("L", [11, 40])       # 40: the q-split formulation (L >= 32)
def test_fused_din_attention_on_a_record_view_without_mask_and_biases(L):
    B, E, H = 200, 16, 64
    q, K, _, state, dout = _case(B, L, E, H, seed=5)
    state[PFX + "attention_layer.mlp.0.bias"].zero_()
    state[PFX + "attention_layer.mlp.2.bias"].zero_()
    ones = torch.ones(B, L, dtype=torch.bool)
    ro, rdq, rdK, rg, _ = _oracle(q, K, ones, state, dout, True)
    mod = _module(E, H, state, fused=True)
    out, dq, dK, grads = _run(mod, q, K, None, dout, True, record_view=True)
    _close(out, ro, 2e-5, "out")
    _close(dq, rdq, 1e-4, "dq")
    _close(dK, rdK, 1e-4, "dK")
    _close(grads["mlp.0.weight"], rg["mlp.0.weight"], 2e-4, "dW1")


@pytest.mark.parametrize("L", [9, 50])
def test_fully_masked_and_fully_kept_histories(L):
    """A sample whose history is all padding pools to zero and sends no pooling gradient into K; the
    attention MLP still sees its positions (Dice statistics run over ALL B*L rows, as in the reference)."""
    B, E, H = 37, 16, 64
    q, K, mask, state, dout = _case(B, L, E, H, seed=21)
    mask[0] = False
    mask[1] = True
    mask[B - 1] = False
    ro, rdq, rdK, rg, _ = _oracle(q, K, mask, state, dout, True)
    mod = _module(E, H, state, fused=True)
    out, dq, dK, grads = _run(mod, q, K, mask, dout, True)
    assert float(out[0].abs().max()) == 0.0 and float(out[B - 1].abs().max()) == 0.0
    _close(out, ro, 2e-5, "out")
    _close(dq, rdq, 1e-4, "dq")
    _close(dK, rdK, 1e-4, "dK")
    _close(grads["mlp.0.weight"], rg["mlp.0.weight"], 2e-4, "dW1")


def test_fused_din_attention_is_deterministic():
    B, L, E, H = 1000, 50, 16, 64
    q, K, mask, state, dout = _case(B, L, E, H, seed=9)
    runs = []
    for _ in range(2):
        mod = _module(E, H, state, fused=True)
        out, dq, dK, grads = _run(mod, q, K, mask, dout, True)
        runs.append([out, dq, dK] + [grads[k] for k in sorted(grads)])
    for x, y in zip(*runs):
        assert torch.equal(x, y)


def test_entry_points_reject_shapes_beyond_the_fused_limits():
    q = torch.zeros(4, 20, device=DEV)
    K = torch.zeros(4, 3, 20, device=DEV)
    W1 = torch.zeros(8, 80, device=DEV)
    with pytest.raises(Exception, match="E <= 16"):
        ops.din_attn_stats(q, K, W1, None, torch.zeros(16, device=DEV), torch.zeros(64, device=DEV))
