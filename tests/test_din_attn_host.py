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
