#!/usr/bin/env python3
"""Headline benchmark: samples/sec of the native DeepFM training step on synthetic Criteo-shaped
input (BASELINE.json configs[1]: 26 sparse + 13 dense fields, 33.76 M rows, emb_dim 16, MLP
4x1024, Adam, batch 4096 per GPU).  One step = pack -> de-dup -> catch-up -> gather -> FM/LR ->
MLP fwd -> sigmoid+BCE -> MLP bwd -> sparse grad reduce -> global-norm clip -> dense + sparse-row
Adam, nothing skipped.  Inputs are resident in HBM before the timed region.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0) carrying `roofline` (dominant kernel: the fp32 MFMA GEMM, timed live
with HIP events inside the timed region) and `cpu_baseline` (the oracle's dense-semantics step —
a restatement of the reference — timed on this box's host cores; rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBS = 8000.0           # HBM3E spec (6.3 TB/s measured streaming ceiling)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4096, help="per-GPU batch")
    ap.add_argument("--model", default="DeepFM", choices=["DeepFM", "DCNv2", "DIN", "DLRM", "xDeepFM"])
    ap.add_argument("--dist", default="powerlaw", choices=["powerlaw", "uniform"])
    ap.add_argument("--sparse-update", default="exact", choices=["exact", "lazy"])
    ap.add_argument("--vocab-scale", type=float, default=1.0,
                    help="scale every table: 3.7 = configs[4]'s 125 M rows (8 GB at D=16)")
    ap.add_argument("--cpu-baseline-steps", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--replicas", action="store_true",
                    help="N > 1: run independent replicas instead of the row-sharded model")
    ap.add_argument("--loader", action="store_true",
                    help="feed the steps from DeviceNpzDataLoader over a synthetic .npz (end-to-end "
                         "rate incl. host batch assembly + H2D); not the headline number")
    ap.add_argument("--host-inputs", action="store_true",
                    help="feed host (DataLoader-style) tensors each step: the PCIe-inclusive rate")
    ap.add_argument("--no-graph", action="store_true",
                    help="launch the step eagerly instead of replaying a captured hipGraph")
    return ap.parse_args()


def build_model(args, device_index, cards, shard=None):
    from fuxictr_amd import synthetic, zoo
    if args.model == "DIN":
        fmap, spec = synthetic.taobao_feature_map(embedding_dim=16, scale=args.vocab_scale)
    else:
        fmap, spec = synthetic.criteo_feature_map(cards=cards, embedding_dim=16)
    common = dict(gpu=device_index, embedding_dim=16, learning_rate=1e-3, optimizer="adam",
                  loss="binary_crossentropy", task="binary_classification",
                  metrics=["logloss", "AUC"], verbose=0, model_root="/tmp/fx_bench",
                  sparse_update=args.sparse_update, hip_graph=not args.no_graph, shard=shard)
    torch.manual_seed(2019)
    if args.model == "DeepFM":
        model = zoo.DeepFM(fmap, model_id="bench", hidden_units=[1024] * 4, **common)
    elif args.model == "DIN":
        model = zoo.DIN(fmap, model_id="bench", dnn_hidden_units=[512, 128, 64],
                        dnn_activations="relu", attention_hidden_units=[64],
                        attention_hidden_activations="Dice", din_target_field=["adgroup_id"],
                        din_sequence_field=["click_sequence"], din_use_softmax=False, **common)
    elif args.model == "DLRM":
        # configs[4]: bottom [512,256] (+ the model's own ->16 layer), top below
        model = zoo.DLRM(fmap, model_id="bench", bottom_mlp_units=[512, 256],
                         top_mlp_units=[1024, 1024, 512, 256], interaction_op="dot", **common)
    elif args.model == "xDeepFM":
        model = zoo.xDeepFM(fmap, model_id="bench", dnn_hidden_units=[1024] * 4,
                            cin_hidden_units=[16, 16, 16], **common)
    else:
        model = zoo.DCNv2(fmap, model_id="bench", model_structure="parallel", num_cross_layers=3,
                          parallel_dnn_hidden_units=[1024] * 4, **common)
    return model, fmap, spec


def _pick_threads():
    """Host threads for the CPU baseline: the fastest of a few counts on a memory-bound pass (what
    the reference's dense Adam is); all visible cores is often NOT the fastest on a big box."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    x = torch.ones(64 << 20)
    best, best_t = 1, None
    for n in sorted({min(avail, c) for c in (8, 16, 32, 64, 128, avail)}):
        torch.set_num_threads(n)
        x.mul_(1.0)
        t0 = time.perf_counter()
        for _ in range(3):
            x.mul_(1.0001)
        dt = time.perf_counter() - t0
        if best_t is None or dt < best_t:
            best, best_t = n, dt
    return best


def cpu_baseline(args, cards, n_steps):
    """The oracle (restatement of the reference, dense [V,D] grads + dense Adam over every row)
    on the host cores, same workload shape, a bounded number of steps."""
    from fuxictr_amd import synthetic
    from oracle import ctr_oracle as O
    torch.set_num_threads(_pick_threads())
    g = torch.Generator().manual_seed(0)
    _, spec = synthetic.criteo_feature_map(cards=cards, embedding_dim=16)
    features = {k: v for item in spec["features"] for k, v in item.items()}
    state = {}
    D = 16
    for f, fs in features.items():
        if fs["type"] == "numeric":
            state[O.EMB + f + ".weight"] = torch.randn(D, 1, generator=g) * 0.3
            state[O.LR_EMB + f + ".weight"] = torch.randn(1, 1, generator=g) * 0.3
        else:
            state[O.EMB + f + ".weight"] = torch.randn(fs["vocab_size"], D, generator=g) * 1e-4
            state[O.LR_EMB + f + ".weight"] = torch.randn(fs["vocab_size"], 1, generator=g) * 1e-4
    state["fm.lr_layer.bias"] = torch.zeros(1)
    dims = [39 * D] + [1024] * 4
    if args.model == "DeepFM":
        for i in range(4):
            state["mlp.mlp.%d.weight" % (2 * i)] = torch.randn(dims[i + 1], dims[i], generator=g) * 0.03
            state["mlp.mlp.%d.bias" % (2 * i)] = torch.zeros(dims[i + 1])
        state["mlp.mlp.8.weight"] = torch.randn(1, 1024, generator=g) * 0.03
        state["mlp.mlp.8.bias"] = torch.zeros(1)
        cfg = {"model": "DeepFM", "n_hidden": 4}
    else:
        for k in list(state):
            if k.startswith("fm."):
                del state[k]
        for i in range(3):
            state["crossnet.cross_layers.%d.weight" % i] = torch.randn(624, 624, generator=g) * 0.03
            state["crossnet.cross_layers.%d.bias" % i] = torch.zeros(624)
        for i in range(4):
            state["parallel_dnn.mlp.%d.weight" % (2 * i)] = torch.randn(dims[i + 1], dims[i], generator=g) * 0.03
            state["parallel_dnn.mlp.%d.bias" % (2 * i)] = torch.zeros(dims[i + 1])
        state["fc.weight"] = torch.randn(1, 624 + 1024, generator=g) * 0.03
        state["fc.bias"] = torch.zeros(1)
        cfg = {"model": "DCNv2", "n_hidden": 4, "n_cross": 3}
    tr = O.OracleTrainer(cfg, state, features, lr=1e-3, max_norm=10.0)
    del state
    rng = np.random.default_rng(1)
    batches = [{k: torch.from_numpy(v) for k, v in
                synthetic.criteo_batch(rng, args.batch, cards=cards, dist=args.dist).items()}
               for _ in range(n_steps + 1)]
    tr.train_step(batches[0], batches[0]["label"])          # warm-up (allocates grads/moments)
    t0 = time.perf_counter()
    done = 0
    for b in batches[1:]:
        tr.train_step(b, b["label"])
        done += 1
        if time.perf_counter() - t0 > 30.0:                  # bounded sample
            break
    dt = time.perf_counter() - t0
    n_steps = done
    return {"value": args.batch * n_steps / dt, "unit": "samples/sec",
            "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d dense-Adam training steps of the oracle (%s, batch %d, full vocab) "
                      "after 1 warm-up, %.1f s" % (n_steps, args.model, args.batch, dt),
            "ms_per_step": 1e3 * dt / n_steps}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus or world == 1, "launch with torch.distributed.run for --gpus > 1"
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    # debug hook for 1-GPU boxes: FX_BENCH_BACKEND=gloo puts every rank on cuda:0 and stages the
    # collectives through the host (RCCL refuses two ranks on one device)
    backend = os.environ.get("FX_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # FX_SHARD_WORLD1=1 (debug, 1-GPU box): a 1-rank process group and the full row-sharded exchange
    # path (all-to-all with itself, all-reduce of one) through RCCL
    world1 = world == 1 and os.environ.get("FX_SHARD_WORLD1") == "1"
    if world1:
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or world1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # a HIP error inside ProcessGroupNCCL's watchdog thread (e.g. hipErrorCapturedEvent from an
        # event query while a hipGraph is being captured — fuxictr_amd/dist.py keeps that from
        # happening) must not take the benchmark down with it
        os.environ.setdefault("TORCH_NCCL_RETHROW_CUDA_ERRORS", "0")
        if backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from fuxictr_amd import ops, synthetic
    from fuxictr_amd.layers import FeatureDict
    cards = [max(3, int(c * args.vocab_scale)) for c in synthetic.CRITEO_CARDS]
    parallelism = "single GPU"
    if (world > 1 or world1) and not args.replicas:
        # row-sharded tables (row % world) + all-to-all exchange + dense all-reduce
        model, fmap, spec = build_model(args, local_rank, cards, shard="row")
        parallelism = ("tables row-sharded over %d ranks (RCCL all-to-all of ids / rows / row "
                       "gradients), towers data-parallel (one flat all-reduce carrying the clip "
                       "norm); %s" % (world, "eager launches" if args.no_graph else
                                      "hipGraph segments with the collectives launched between them"))
    else:
        model, fmap, spec = build_model(args, local_rank, cards)
        if world > 1:
            parallelism = "%d independent replicas (--replicas; no data-path collective)" % world
    model.train()

    # synthetic batches, resident in HBM (ids int64 / dense fp32 / label fp32 — what the
    # reference's get_inputs would hold after .to(device)); distinct per rank and per step
    rng = np.random.default_rng(1000 + rank)
    n_pool = 8
    pool = []
    for _ in range(n_pool):
        if args.model == "DIN":
            b = synthetic.taobao_batch(rng, args.batch, spec, dist=args.dist)
        else:
            b = synthetic.criteo_batch(rng, args.batch, cards=cards, dist=args.dist)
        if args.host_inputs:
            # what the reference's DataLoader yields: host tensors, int64 ids / float64 numerics
            pool.append({k: torch.from_numpy(v.astype(np.float64) if v.dtype == np.float32 else v)
                         for k, v in b.items()})
        else:
            pool.append({k: torch.from_numpy(v).to(dev) for k, v in b.items()})

    loader_iter = None
    if args.loader:
        from fuxictr_amd.dataloader import DeviceNpzDataLoader
        n_rows = args.batch * 64
        rng2 = np.random.default_rng(7 + rank)
        big = synthetic.taobao_batch(rng2, n_rows, spec, dist=args.dist) if args.model == "DIN" \
            else synthetic.criteo_batch(rng2, n_rows, cards=cards, dist=args.dist)
        npz_path = "/tmp/fx_bench_loader_%d.npz" % rank
        np.savez(npz_path, **big)
        del big

        loader = DeviceNpzDataLoader(fmap, npz_path, batch_size=args.batch, shuffle=True,
                                     device=dev, seed=rank)

        def _batches():
            while True:                                   # epochs
                for bt in loader:
                    if bt[fmap.labels[0]].shape[0] == args.batch:
                        yield bt
        loader_iter = _batches()

    def next_batch(i):
        return next(loader_iter) if loader_iter is not None else pool[i % n_pool]

    def sync():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    step_i = 0
    launch_note = None
    try:
        for _ in range(max(args.warmup, 5 if model._use_graph else 0)):   # >= 5: 3 eager + capture
            model.train_step(next_batch(step_i))
            step_i += 1
        sync()
    except Exception as exc:   # noqa: BLE001 — e.g. a capture problem on a software stack not seen
        if not model._use_graph:                      # in development: keep the run, launch eagerly
            raise
        launch_note = "hipGraph capture failed (%s: %s); eager launches" % (type(exc).__name__, exc)
        print("[bench] " + launch_note, file=sys.stderr, flush=True)
        model._use_graph = False
        model._graph_state = None
        for _ in range(args.warmup):
            model.train_step(pool[step_i % n_pool])
            step_i += 1
        sync()
    ops.KernelTimer.reset()
    # eager mode: the roofline kernels are timed with HIP events inside the timed region itself;
    # graph mode: events cannot sit inside a replayed graph, so the same kernels are timed in an
    # instrumented eager pass right after the timed region (same process, same buffers)
    ops.KernelTimer.enabled = (not model._use_graph) and not args.no_kernel_timing
    t0 = time.perf_counter()
    for _ in range(args.steps):
        model.train_step(next_batch(step_i))
        step_i += 1
    sync()
    dt = time.perf_counter() - t0
    ops.KernelTimer.enabled = False
    model.optimizer.check_errors()
    timing_mode = "hip events inside the timed region (eager launches)"
    if model._use_graph and not args.no_kernel_timing:
        # the eager pass runs on the default stream; the parameters' AccumulateGrad nodes were
        # created on the capture stream — harmless here, silence the per-parameter warning
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        model._use_graph = False
        ops.KernelTimer.enabled = True
        for _ in range(min(args.steps, 20)):
            model.train_step(pool[step_i % n_pool])
            step_i += 1
        sync()
        ops.KernelTimer.enabled = False
        model._use_graph = True
        timing_mode = ("hip events around the same kernels in an eager pass of %d steps right "
                       "after the timed region (the timed region replays a hipGraph)"
                       % min(args.steps, 20))
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ktimes = ops.KernelTimer.summary()

    if rank == 0:
        global_batch = args.batch * world
        value = global_batch * args.steps / dt
        out = {
            "metric": "samples/sec at batch 4096, Criteo-shape DeepFM/DCNv2, 1/2/4/8 MI355X",
            "value": value, "unit": "samples/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": ("configs[3]: DIN on synthetic Taobao-shape sequences (14 "
                                    "categorical + click_sequence len 50 sharing adgroup_id, "
                                    "emb_dim 16, attention [64] Dice, dnn [512,128,64]), Adam, "
                                    "full training step") if args.model == "DIN" else
                                   "configs[%d]: %s on synthetic Criteo (26 sparse + 13 dense, "
                                   "%d rows, emb_dim 16, %s), Adam, full training step"
                                   % ({"DeepFM": 1, "DCNv2": 2, "DLRM": 4}.get(args.model, 1),
                                      args.model, sum(cards) + len(cards),
                                      {"DCNv2": "MLP 4x1024, 3 cross layers",
                                       "DLRM": "bottom MLP [512,256,16], dot interaction, "
                                               "top MLP [1024,1024,512,256]",
                                       "xDeepFM": "MLP 4x1024, CIN [16,16,16]"}.get(
                                          args.model, "MLP 4x1024")),
                       "global_batch": global_batch, "per_gpu_batch": args.batch,
                       "id_distribution": args.dist, "sparse_update": args.sparse_update,
                       "launch": launch_note or ("hipGraph replay" if model._use_graph else "eager"),
                       "inputs": ("host tensors per step (DataLoader-style; one pinned staging "
                                  "copy per dtype) - PCIe-inclusive, NOT the headline number")
                       if args.host_inputs else
                       ("DeviceNpzDataLoader over a 64-batch synthetic .npz, shuffled (host column "
                        "gather + pinned H2D on a copy stream, prefetched) - end-to-end, NOT the "
                        "headline number") if args.loader else "resident in HBM",
                       "parallelism": parallelism},
        }
        g = ktimes.get("k_gemm_f32")
        traffic = None
        if args.model == "DeepFM" and args.batch == 4096 and world == 1:
            # fabric-side bytes per launch from the committed PMC passes (rocprofv3 --pmc cannot run
            # inside this process); only valid for the exact workload it was collected on
            try:
                with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles",
                                       "r01_pmc_traffic.json")) as f:
                    traffic = json.load(f)["traffic_bytes_per_launch"]
            except (OSError, ValueError, KeyError):
                traffic = None
        if g and g["total_ms"] > 0:
            ach = g["work"] / (g["total_ms"] * 1e-3) / 1e12
            out["roofline"] = {"kernel": "k_gemm_f32 (fp32 MFMA GEMM, MLP/CrossNet fwd+bwd)",
                               "bound": "mfma", "achieved": ach, "peak": PEAK_FP32_MFMA_TFLOPS,
                               "unit": "TFLOP/s", "frac": ach / PEAK_FP32_MFMA_TFLOPS,
                               "traffic": traffic, "traffic_unit": "bytes per launch (L2 fabric "
                               "requests incl. Infinity-Cache hits; profiles/r01_pmc_traffic.txt)",
                               "launches": g["launches"],
                               "avg_launch_us": g["avg_us"], "timing": timing_mode,
                               "gemm_share_of_instrumented_step": None}
        e = ktimes.get("k_emb_gather_fwd")
        if e and e["total_ms"] > 0:
            ach = e["work"] / (e["total_ms"] * 1e-3) / 1e9
            out["roofline_gather"] = {"kernel": "k_emb_gather_fwd", "bound": "hbm", "achieved": ach,
                                      "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                      "frac": ach / PEAK_HBM_GBS, "traffic": None,
                                      "launches": e["launches"], "avg_launch_us": e["avg_us"]}
        if world == 1 and not args.no_cpu_baseline and args.model in ("DeepFM", "DCNv2"):
            del pool
            out["cpu_baseline"] = cpu_baseline(args, cards, args.cpu_baseline_steps)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
