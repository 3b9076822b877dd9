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
