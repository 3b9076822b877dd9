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


def _bench_line(args, env=None, timeout=900):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, capture_output=True,
                       text=True, env=e, timeout=timeout)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]            # ONE JSON line, printed by rank 0
    return json.loads(lines[0])


def test_bench_launch_with_two_ranks_reports_two_gpus_and_the_one_rank_losses():
    """the launch path the driver uses for N > 1 (`bench.py --gpus N` -> torch.distributed.run
    -> one rank per device -> row-sharded tables, all-to-alls, one all-reduce, hipGraph segments), here with
    both ranks on cuda:0 and the collectives staged through gloo.  The line must say n_gpus = 2, name the
    parallelism, scale the global batch, and — same seeded model, same seeded GLOBAL batch — reproduce the
    losses of the one-rank run."""
    common = ["--vocab-scale", "0.01", "--steps", "3", "--warmup", "5", "--no-cpu-baseline",
              "--no-kernel-timing", "--no-dcnv2", "--probe-loss"]
    one = _bench_line(["--gpus", "1", "--probe-world", "2"] + common)
    two = _bench_line(["--gpus", "2"] + common, env={"FX_BENCH_BACKEND": "gloo"})
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["config"]["global_batch"] == 2 * two["config"]["per_gpu_batch"] == 8192
    assert "row-sharded" in two["config"]["parallelism"] and "all-to-all" in two["config"]["parallelism"]
    assert two["scaling"] == "weak" and two["value"] > 0
    assert len(one["probe_loss"]) == len(two["probe_loss"]) == 2
    for a, b in zip(one["probe_loss"], two["probe_loss"]):
        assert abs(a - b) <= 2e-5, (one["probe_loss"], two["probe_loss"])
    assert one["probe_loss"][0] != one["probe_loss"][1]          # the model did move between the steps


def test_collectives_recorded_into_the_graph_run_and_exit_on_one_rccl_rank():
    """Round-4 default on RCCL: the sharded step as ONE hipGraph with the collectives recorded in it
    (FX_GRAPH_COLLECTIVES=0 = the hipGraph segments with eager collectives between them).  Rounds 2 / 3 hung at
    teardown — RCCL's communicator waits for every graph that recorded its kernels — so the captured step
    is released before destroy_process_group (DistContext.shutdown).  One rank, real RCCL: the line comes
    out, the process ends by itself with exit code 0, the line says which form ran, and the probe losses
    equal those of the segmented step (same kernels, same order)."""
    common = ["--vocab-scale", "0.01", "--steps", "3", "--warmup", "5", "--no-cpu-baseline",
              "--no-kernel-timing", "--no-dcnv2", "--probe-loss"]
    seg = _bench_line(common, env={"FX_SHARD_WORLD1": "1", "FX_GRAPH_COLLECTIVES": "0"})
    rec = _bench_line(common, env={"FX_SHARD_WORLD1": "1"}, timeout=240)
    assert "recorded" in rec["config"]["parallelism"] and "segments" in seg["config"]["parallelism"]
    assert rec["probe_loss"] == seg["probe_loss"]


def test_a_failed_capture_of_the_collectives_falls_back_to_segments():
    """The recorded-collectives form is the default; a stack on which recording an RCCL kernel throws must
    not take the run down: the step falls back to hipGraph segments with eager collectives by itself, the
    line says so, and the probe losses are those of the recorded form."""
    common = ["--vocab-scale", "0.01", "--steps", "3", "--warmup", "5", "--no-cpu-baseline",
              "--no-kernel-timing", "--no-dcnv2", "--probe-loss"]
    rec = _bench_line(common, env={"FX_SHARD_WORLD1": "1"}, timeout=240)
    fb = _bench_line(common, env={"FX_SHARD_WORLD1": "1", "FX_TEST_CAPTURE_FAIL": "1"}, timeout=240)
    assert "recorded" in rec["config"]["parallelism"]
    assert "segments" in fb["config"]["parallelism"], fb["config"]["parallelism"]
    assert fb["probe_loss"] == rec["probe_loss"]
    assert fb["value"] > 0


@pytest.mark.parametrize("dist", ["powerlaw", "uniform"])
def test_eight_ranks_on_one_gpu_at_the_real_c5_shapes(dist):
    """the first real 8-GPU run, rehearsed on the one GPU there is.  configs[4] as BASELINE.json
    states it — DLRM bottom [512,256] / top [1024,1024,512,256], tables scaled x 3.7 (124.9 M rows: 8 GB of
    embeddings + 16 GB of Adam state over the 8 ranks), global batch 32 768 — with EIGHT ranks sharing cuda:0 (the
    real kernels, the collectives staged through gloo: RCCL refuses two ranks on one device).  Same seeded model
    (weights by parameter name and global row), same seeded global batch: the two probe losses equal those of the
    one-rank run on the whole batch, the per-owner buckets of the all-to-all never overflow (the run would abort
    through optimizer.check_errors) and their fullest fill against the fixed 1.5 x capacity is printed — power-law
    and uniform ids."""
    common = ["--model", "DLRM", "--vocab-scale", "3.7", "--dist", dist, "--steps", "2", "--warmup", "3",
              "--age-steps", "0", "--no-cpu-baseline", "--no-kernel-timing", "--no-step-events", "--no-parity",
              "--no-uniform", "--probe-loss"]
    one = _bench_line(["--gpus", "1", "--probe-world", "8"] + common, timeout=1200)
    eight = _bench_line(["--gpus", "8"] + common, env={"FX_BENCH_BACKEND": "gloo"}, timeout=1500)
    assert eight["n_gpus"] == 8 and eight["config"]["global_batch"] == 32768
    assert "row-sharded over 8 ranks" in eight["config"]["parallelism"]
    for a, b in zip(one["probe_loss"], eight["probe_loss"]):
        assert abs(a - b) <= 2e-5, (one["probe_loss"], eight["probe_loss"])
    fill = eight["a2a_bucket"]
    print("[8 ranks, c5 shapes, %s] probe losses %s | fullest all-to-all bucket %d of %d rows (%.2f of capacity; "
          "even share of the lookups %d)" % (dist, eight["probe_loss"], fill["fullest_bucket_rows"],
                                             fill["capacity_rows"], fill["fill"], fill["even_share_of_lookups"]))
    assert 0.0 < fill["fill"] < 1.0


def test_bf16_tables_row_sharded_over_two_ranks_reproduce_the_one_rank_losses():
    """Round 6: `emb_dtype: bf16` with `shard: row` (it raised NotImplementedError until
    now).  The owner widens its bf16 rows — after the rounding their stored copy went through — into the fp32
    block of the exchange; requesters read them as an unsharded bf16 table's rows are read; moments, gradients
    and the update arithmetic are fp32 on the owner, the updated row is rounded to nearest-even bf16.  Same
    seeded model, same seeded GLOBAL batch: the two probe losses of the 2-rank run (both ranks on cuda:0,
    collectives staged through gloo) equal those of the one-rank bf16 run on the whole batch."""
    common = ["--vocab-scale", "0.01", "--steps", "3", "--warmup", "5", "--no-cpu-baseline", "--no-kernel-timing",
              "--no-dcnv2", "--no-din", "--no-parity", "--no-uniform", "--probe-loss", "--emb-dtype", "bf16"]
    one = _bench_line(["--gpus", "1", "--probe-world", "2"] + common)
    two = _bench_line(["--gpus", "2"] + common, env={"FX_BENCH_BACKEND": "gloo"})
    assert two["n_gpus"] == 2 and "row-sharded" in two["config"]["parallelism"]
    assert "bf16" in two["config"]["emb_dtype"]
    for a, b in zip(one["probe_loss"], two["probe_loss"]):
        assert abs(a - b) <= 2e-5, (one["probe_loss"], two["probe_loss"])
    assert one["probe_loss"][0] != one["probe_loss"][1]
