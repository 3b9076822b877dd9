"""Row-sharded path with the REAL kernels on a 1-GPU box: two ranks share cuda:0 and the
collectives are staged through gloo (RCCL refuses two ranks on one device), which exercises
everything of the multi-GPU step except RCCL itself.  Same oracle as tests/test_dist_gloo.py: the
reference's single-process run on the full batches."""
import pytest

pytestmark = pytest.mark.gpu

from conftest import Golden  # noqa: E402
from test_dist_gloo import check_against_golden, run_workers  # noqa: E402


@pytest.mark.parametrize("case", ["deepfm_adam", "dcnv2_adam"])
def test_two_ranks_one_gpu_equals_reference(case, tmp_path):
    g = Golden(case)
    z = run_workers(case, tmp_path, use_gpu=True)
    check_against_golden(z, g)


@pytest.mark.parametrize("case", ["deepfm_adam", "dcnv2_adam"])
def test_two_ranks_segmented_hip_graph_equals_reference(case, tmp_path):
    """hip_graph + shard: the step is replayed as hipGraph segments with the collectives launched
    eagerly in between (3 eager warm-up steps, capture on the 4th, replays after that)."""
    g = Golden(case)
    assert g.meta["steps"] >= 5
    z = run_workers(case, tmp_path, use_gpu=True, env={"FX_HIP_GRAPH": "1"})
    check_against_golden(z, g)


@pytest.mark.parametrize("graph", ["0", "1"])
@pytest.mark.parametrize("case", ["deepfm_adam", "xdeepfm_adam"])
def test_rccl_backend_one_rank_full_exchange_path(case, graph, tmp_path):
    """The `nccl` (= RCCL) backend itself: a 1-rank group with FX_SHARD_WORLD1=1 keeps the whole
    row-sharded step — int32 / fp32 all_to_all_single, the flat all-reduce, eager collectives
    between hipGraph segments — on RCCL's streams (what the 2-rank gloo tests cannot cover)."""
    g = Golden(case)
    z = run_workers(case, tmp_path, use_gpu=True, world=1,
                    env={"FX_SHARD_WORLD1": "1", "FX_TEST_BACKEND": "nccl", "FX_HIP_GRAPH": graph})
    assert bool(z["sharded"][0])
    check_against_golden(z, g)
