"""Row-sharded multi-process path, 2 ranks over gloo in the GPU-less container (kernels = test
emulation): ids all-to-all -> owner gather -> rows all-to-all -> ... -> gradient all-to-all -> owner
update, dense all-reduce, global clip.  Each rank trains on half of every golden batch; the result
must equal the REFERENCE's single-process run on the full batches (tests/golden)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import Golden, assert_weights_close

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def run_workers(case, tmp_path, use_gpu, world=2, env=None):
    port = _free_port()
    out = str(tmp_path / "out.npz")
    e = dict(os.environ)
    e.update(env or {})
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), str(r),
                               str(world), case, str(port), out, "1" if use_gpu else "0"], env=e)
             for r in range(world)]
    codes = [p.wait(timeout=900) for p in procs]
    assert codes == [0] * world, codes
    return np.load(out)


def check_against_golden(z, g):
    np.testing.assert_allclose(z["pred0"], g.expect["pred0"], atol=2e-6)
    np.testing.assert_allclose(z["losses"], g.expect["loss"], atol=1e-5)
    np.testing.assert_allclose(z["pred1"], g.expect["pred1"], atol=2e-5)
    for k, ref in g.state1.items():
        assert_weights_close(z["state/" + k], ref, g.meta["lr"], g.meta["steps"], k)


# (DIN: Dice normalises with the statistics of the WHOLE batch — its column sums are all-reduced
# across the ranks, layers._DiceFn — so the 2-rank run is the reference's 1-rank trajectory too)
@pytest.mark.parametrize("case", ["deepfm_adam", "dcnv2_adam", "deepfm_adam_clip", "dlrm_adam",
                                  "xdeepfm_adam", "deepfm_seqpool", "dcnv2_mixdim", "din_adam",
                                  "din_pairs_softmax", "deepfm_reg", "deepfm_reg_sgd"])
def test_two_rank_sharded_training_equals_reference(case, tmp_path):
    g = Golden(case)
    z = run_workers(case, tmp_path, use_gpu=False)
    check_against_golden(z, g)


def test_three_ranks_uneven_shards(tmp_path):
    g = Golden("deepfm_sgd")
    z = run_workers("deepfm_sgd", tmp_path, use_gpu=False, world=3)
    check_against_golden(z, g)


def test_sharded_checkpoint_files_roundtrip(tmp_path):
    """save_weights / load_weights with row-sharded tables: one file per rank."""
    g = Golden("deepfm_adam")
    z = run_workers("deepfm_adam", tmp_path, use_gpu=False, env={"FX_TEST_CKPT": "1"})
    check_against_golden(z, g)


def test_one_rank_group_runs_the_full_exchange_path(tmp_path):
    """FX_SHARD_WORLD1=1: a 1-rank process group keeps the sharded path (all-to-all with itself) —
    the hook that runs the RCCL code path end to end on a 1-GPU box."""
    g = Golden("deepfm_adam")
    z = run_workers("deepfm_adam", tmp_path, use_gpu=False, world=1, env={"FX_SHARD_WORLD1": "1"})
    assert bool(z["sharded"][0])
    check_against_golden(z, g)


def test_fit_takes_the_same_decisions_on_every_rank(tmp_path):
    """BaseModel.fit under shard='row' (ADVICE r1): each rank evaluates the GLOBAL validation set
    (shard predictions gathered), so best metric, lr, step count, stop flag and epoch agree on all
    ranks, and the metric equals scikit-learn's on the gathered predictions."""
    from sklearn.metrics import log_loss, roc_auc_score
    g = Golden("deepfm_adam")
    z = run_workers("deepfm_adam", tmp_path, use_gpu=False, env={"FX_TEST_FIT": "1"})
    fit = z["fit"]
    assert fit.shape[0] == 2
    np.testing.assert_array_equal(fit[0], fit[1])
    y = np.asarray(g.batches[-1]["label"], dtype=np.float64)
    assert abs(fit[0, 5] - roc_auc_score(y, z["pred"])) < 1e-9
    assert abs(fit[0, 6] - log_loss(y, z["pred"])) < 1e-6


def test_c5_shaped_dlrm_sharded_equals_the_oracle(tmp_path):
    """BASELINE configs[4] in miniature (SURVEY.md 8e parity test): the c5 DLRM at its real layer
    widths, tables row-sharded over 2 ranks, each rank on its half of the global batch == the
    oracle's single-process run on the full batches."""
    z = run_workers("c5_dlrm", tmp_path, use_gpu=False)
    np.testing.assert_allclose(z["losses"], z["ref_losses"], atol=1e-5)
    np.testing.assert_allclose(z["pred"], z["ref_pred"], atol=2e-5)
    assert float(z["wdiff"][0]) <= 1e-3 * 4 + 2e-5       # (Adam's eps-conditioning, see conftest)


@pytest.mark.parametrize("case", ["deepfm_adam_clip", "dlrm_adam"])
def test_eight_ranks_like_c5(case, tmp_path):
    """BASELINE configs[4]'s world size (8 ranks, owner = row % 8): every rank owns an eighth of
    every table and trains on an eighth of each golden batch — routing, caps and the global clip at
    the world size the N = 8 line of the driver runs at."""
    g = Golden(case)
    z = run_workers(case, tmp_path, use_gpu=False, world=8)
    check_against_golden(z, g)
