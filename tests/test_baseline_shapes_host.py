"""CPU: the baseline-shape parity harness (tests/baseline_shapes.py) itself, with the kernels replaced
by the test-only torch emulation — real layer widths, reduced batch and tables — so that the
`-m gpu` run of tests/test_gpu_baseline_shapes.py is not the first time its logic executes, and the
host wiring (plans, autograd nodes, exact-mode step protocol) is checked at F = 39 / L = 50 shapes."""
import pytest

import _cpu_emul
import baseline_shapes as BS


@pytest.mark.parametrize("case,dist", [("c2_deepfm", "powerlaw"), ("c3_dcnv2", "uniform"),
                                       ("c4_din", "powerlaw"), ("c5_dlrm", "uniform")])
def test_harness_on_the_emulation(case, dist, tmp_path, monkeypatch):
    _cpu_emul.install(monkeypatch)
    from fuxictr_amd import optim, zoo
    from oracle import ctr_oracle as O
    monkeypatch.setattr(optim._NativeOptimizer, "_require_cuda", False)
    model, features, cfg, spec, cards = BS.build(case, zoo, -1, tmp_path, vocab_scale=0.002)
    res = BS.run_parity(case, dist, model, features, cfg, spec, cards, O, B=512, steps=6,
                        holdout=8192)
    assert res["loss_first_last"][1] < res["loss_first_last"][0] + 0.05
    assert res["native"]["max"] <= res["bounds"]["max"]
