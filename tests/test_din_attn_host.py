"""CPU check of the fused DIN attention's HOST side (layers._DinAttnFn: pass order, statistics
hand-over, gradient routing, running statistics) with the kernels replaced by the test-only torch
emulation — against the oracle's restatement of target_attention.py:66-92 + activations.py:40-51 in
fp64.  The kernels themselves are checked on the MI355X by tests/test_gpu_din_attn.py, whose case
builders are reused here."""
import pytest
import torch

import _cpu_emul
import test_gpu_din_attn as G


@pytest.fixture
def emulated(monkeypatch):
    _cpu_emul.install(monkeypatch)
    import fuxictr_amd.layers as nat
    monkeypatch.setattr(nat, "_DEFAULT_DEVICE", torch.device("cpu"))
    monkeypatch.setattr(G, "DEV", "cpu")


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("B,L,E,H", [(64, 50, 16, 64), (7, 3, 4, 16), (33, 40, 8, 36), (5, 1, 10, 64)])
def test_fused_din_attention_host_wiring_matches_the_oracle(B, L, E, H, training, emulated):
    G.test_fused_din_attention_matches_the_oracle(B, L, E, H, training)


def test_fused_and_unfused_host_paths_agree(emulated):
    G.test_fused_and_unfused_native_paths_agree(40, 6, 8, 16)


def test_record_view_no_mask(emulated):
    G.test_fused_din_attention_on_a_record_view_without_mask_and_biases(11)


def test_mlp_tower_pads_an_unaligned_input_width(emulated):
    """layers._MLPFn: an input width that is not a multiple of 4 floats (DLRM: 367) is zero-padded for
    the first layer's GEMMs; outputs and every gradient equal the plain torch stack."""
    import fuxictr_amd.layers as nat
    g = torch.Generator().manual_seed(0)
    B, K, H1, H2 = 33, 367, 40, 8
    x = torch.randn(B, K, generator=g).requires_grad_(True)
    W0, b0 = torch.randn(H1, K, generator=g) * 0.1, torch.randn(H1, generator=g)
    W1, b1 = torch.randn(H2, H1, generator=g) * 0.1, torch.randn(H2, generator=g)
    ps = [t.clone().requires_grad_(True) for t in (W0, b0, W1, b1)]
    y = nat._MLPFn.apply(x, (True, False), None, None, None, None, *ps)
    gy = torch.randn(B, H2, generator=g)
    y.backward(gy)
    xr = x.detach().clone().requires_grad_(True)
    pr = [t.clone().requires_grad_(True) for t in (W0, b0, W1, b1)]
    yr = torch.relu(xr @ pr[0].t() + pr[1]) @ pr[2].t() + pr[3]
    yr.backward(gy)
    assert torch.allclose(y, yr, atol=1e-5)
    assert x.grad.shape == (B, K) and torch.allclose(x.grad, xr.grad, atol=1e-5)
    for a, b in zip(ps, pr):
        assert a.grad.shape == b.grad.shape and torch.allclose(a.grad, b.grad, atol=1e-4)
