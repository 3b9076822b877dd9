#!/usr/bin/env python3
"""Headline benchmark: samples/sec of the native DeepFM training step on synthetic Criteo-shaped
input (BASELINE.json configs[1]: 26 sparse + 13 dense fields, 33.76 M rows, emb_dim 16, MLP
4x1024, Adam, batch 4096 per GPU).  One step = pack -> de-dup -> catch-up -> gather -> FM/LR ->
MLP fwd -> sigmoid+BCE -> MLP bwd -> sparse grad reduce -> global-norm clip -> dense + sparse-row
Adam, nothing skipped.  Inputs are resident in HBM before the timed region.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` launched plainly starts N ranks itself (torch.distributed.run, 127.0.0.1).

Prints ONE JSON line (rank 0) carrying `roofline` (dominant kernel: the fp32 MFMA GEMM, timed live
with HIP events), `roofline_sparse` (the whole embedding path of the step against SURVEY 8d's bytes),
`dcnv2` (the configs[2] step measured in the same run: the metric names both models) and
`cpu_baseline` (the oracle's dense-semantics step — a restatement of the reference — timed on this
box's host cores; rank 0, N=1 only).
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
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16)
PEAK_HBM_GBS = 8000.0           # HBM3E spec (6.3 TB/s measured streaming ceiling)
# The tower / CrossNet GEMMs run on the split-bf16 kernels (fuxictr_amd/csrc/fx_gemm_x6.hip, round 5): six bf16
# MFMA products per fp32 product, so the matrix-core peak for the ALGORITHMIC (fp32) flops is a sixth of the
# bf16 peak.  FX_GEMM_BF16X6=0 restores the fp32-MFMA kernels (peak 157.3).
X6_ON = os.environ.get("FX_GEMM_BF16X6", "1") != "0"
PEAK_GEMM_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0 if X6_ON else PEAK_FP32_MFMA_TFLOPS


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4096, help="per-GPU batch")
    ap.add_argument("--model", default="DeepFM", choices=["DeepFM", "DCNv2", "DIN", "DLRM", "xDeepFM"])
    ap.add_argument("--dist", default="powerlaw", choices=["powerlaw", "uniform"])
    ap.add_argument("--sparse-update", default="exact", choices=["exact", "lazy"])
    ap.add_argument("--emb-dtype", default="fp32", choices=["fp32", "bf16"],
                    help="storage of the embedding tables (bf16: opt-in, NOT the headline number — "
                         "the reference computes and stores fp32)")
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
    ap.add_argument("--pool", type=int, default=512,
                    help="distinct synthetic batches cycled (bounded by warmup + steps + 24)")
    ap.add_argument("--no-dcnv2", action="store_true",
                    help="skip the DCNv2 (configs[2]) sub-measurement of the default DeepFM run")
    ap.add_argument("--probe-loss", action="store_true",
                    help="before the warm-up: seeded weights (by parameter NAME and global row, so any "
                         "shard layout holds the same model) and two training steps on a seeded GLOBAL "
                         "batch; their losses go into the line as probe_loss and must not depend on the "
                         "number of ranks (tests/test_gpu_dist.py)")
    ap.add_argument("--probe-world", type=int, default=0,
                    help="with --probe-loss on fewer ranks: the world size whose global batch to probe")
    ap.add_argument("--zoo", default="native", choices=["native", "reference"],
                    help="reference: time the reference's OWN unmodified model_zoo classes behind "
                         "fuxictr_amd.patch.install() (needs a reference checkout: FX_REFERENCE_ROOT, "
                         "default <repo>/.ref_checkout or /root/reference) - the true drop-in; native: "
                         "fuxictr_amd.zoo's mirrors of the same classes (what travels to the GPU box)")
    ap.add_argument("--no-step-events", action="store_true",
                    help="skip the second pass that records one HIP event per step (step_us)")
    ap.add_argument("--no-graph", action="store_true",
                    help="launch the step eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--age-steps", type=int, default=300,
                    help="training steps (on distinct batches) before the clock starts: the exact-mode Adam "
                         "catch-up replays the steps a row missed, so a young run is faster than the rate a "
                         "training run sustains (round 4: 0.981 vs 1.035-1.046 ms); `value` is the steady state, "
                         "the young-run figure is reported beside it (`young_run`)")
    ap.add_argument("--no-uniform", action="store_true",
                    help="skip the uniform-id measurement (`value_uniform`, SURVEY 8d: both distributions)")
    ap.add_argument("--no-din", action="store_true",
                    help="skip the DIN (configs[3]) sub-measurement of the default DeepFM run")
    ap.add_argument("--long-parity-steps", type=int, default=400,
                    help="steps of the long native-vs-oracle trajectory (`parity_full_vocab.steps_400`: c2 with "
                         "the tables x 0.01, 131 072 hold-out rows; ~1 min); 0 skips it")
    ap.add_argument("--no-parity", action="store_true",
                    help="skip the full-vocabulary parity leg (`parity_full_vocab`: the native step against the "
                         "oracle at 33.76 M rows, same weights, same batches)")
    return ap.parse_args()


def _reference_zoo():
    """The reference's own model_zoo classes on the native layers."""
    import types
    root = os.environ.get("FX_REFERENCE_ROOT")
    if not root:
        for cand in (os.path.join(ROOT, ".ref_checkout"), "/root/reference"):
            if os.path.isdir(os.path.join(cand, "fuxictr")):
                root = cand
                break
    if not root or not os.path.isdir(os.path.join(root, "fuxictr")):
        sys.exit("bench.py --zoo reference: no reference checkout (set FX_REFERENCE_ROOT)")
    for name in ["polars", "h5py", "keras_preprocessing", "keras_preprocessing.sequence"]:
        sys.modules.setdefault(name, types.ModuleType(name))     # import-time only
    sys.modules["keras_preprocessing.sequence"].pad_sequences = lambda *a, **k: None
    sys.dont_write_bytecode = True
    if root not in sys.path:
        sys.path.insert(0, root)
    from fuxictr_amd import patch
    patch.install()
    import model_zoo
    from model_zoo.DeepFM.DeepFM_torch.src import DeepFM as RefDeepFM
    return types.SimpleNamespace(DeepFM=RefDeepFM, DCNv2=model_zoo.DCNv2, DIN=model_zoo.DIN,
                                 DLRM=model_zoo.DLRM, xDeepFM=model_zoo.xDeepFM)


def build_model(args, device_index, cards, shard=None):
    from fuxictr_amd import synthetic, zoo
    if args.zoo == "reference":
        zoo = _reference_zoo()
    if args.model == "DIN":
        fmap, spec = synthetic.taobao_feature_map(embedding_dim=16, scale=args.vocab_scale)
    else:
        fmap, spec = synthetic.criteo_feature_map(cards=cards, embedding_dim=16)
    common = dict(gpu=device_index, embedding_dim=16, learning_rate=1e-3, optimizer="adam",
                  loss="binary_crossentropy", task="binary_classification",
                  metrics=["logloss", "AUC"], verbose=0, model_root="/tmp/fx_bench",
                  sparse_update=args.sparse_update, hip_graph=not args.no_graph, shard=shard,
                  emb_dtype=args.emb_dtype)
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


def _host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


# what the survey measured with the REAL, unmodified reference on 8 host cores of the build
# container (it cannot run on the GPU box: /root/reference does not travel)
REFERENCE_MEASURED = {"DeepFM": {"value": 2069.0, "ms_per_step": 1980.0},
                      "DCNv2": {"value": 2169.0, "ms_per_step": 1889.0},
                      "DIN": {"value": 8654.0, "ms_per_step": 473.0}}


def cpu_baseline(args, cards, n_steps):
    """The oracle (restatement of the reference, dense [V,D] grads + dense Adam over every row)
    on the host cores, same workload shape, a bounded number of steps.  The thread count is
    calibrated on the STEP itself (one step per candidate count; all visible cores is usually not
    the fastest for a step that is a 16 GB memory stream)."""
    from fuxictr_amd import synthetic
    from oracle import ctr_oracle as O
    avail = _host_cores()
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

    def batch():
        return {k: torch.from_numpy(v) for k, v in
                synthetic.criteo_batch(rng, args.batch, cards=cards, dist=args.dist).items()}

    def one_step():
        b = batch()
        t0 = time.perf_counter()
        tr.train_step(b, b["label"])
        return time.perf_counter() - t0
    torch.set_num_threads(min(avail, 32))
    one_step()                                              # warm-up (allocates grads/moments)
    t_begin = time.perf_counter()
    best, best_t, tried = None, None, {}
    for n in sorted({min(avail, c) for c in (8, 16, 32, 64, 128)}):
        torch.set_num_threads(n)
        dt = one_step()
        tried[n] = round(1e3 * dt, 1)
        if best_t is None or dt < best_t:
            best, best_t = n, dt
        # bounded: stop once more threads make it clearly slower (256 threads: 56 s per step measured)
        if dt > 1.3 * best_t or time.perf_counter() - t_begin > 15.0:
            break
    torch.set_num_threads(best)
    t0 = time.perf_counter()
    done = 0
    while done < n_steps or (done < 3):
        one_step_dt = one_step()
        done += 1
        if time.perf_counter() - t0 > 20.0:                  # bounded sample
            break
    dt = time.perf_counter() - t0
    # batch generation (numpy, ~2 ms) is inside dt: < 0.2 % of a ~1.5 s step
    return {"value": args.batch * done / dt, "unit": "samples/sec",
            "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d dense-Adam training steps of the oracle (%s, batch %d, full vocab) after "
                      "1 warm-up and a thread-count calibration on the step itself (ms per step by "
                      "threads: %s), %.1f s" % (done, args.model, args.batch, tried, dt),
            "ms_per_step": 1e3 * dt / done, "host_cores": avail,
            "reference_measured": dict(REFERENCE_MEASURED.get(args.model, {}),
                                       note="the real reference, 8 cores of the build container "
                                            "(BASELINE.md section 2); not re-measurable on the GPU "
                                            "box")}


def reference_dataloader_rate(args, batches=24):
    """SURVEY.md 8d "a second number *with* the reference DataLoader": samples/s the reference's OWN
    RankDataLoader -> NpzDataLoader (rank_dataloader.py:84-101, npz_dataloader.py:35-66: per-row __getitem__,
    default_collate, BatchCollator) hands out on this host for the c2 Criteo shape — the ceiling of ANY model
    behind it, native or not.  Needs a reference checkout (FX_REFERENCE_ROOT, <repo>/.ref_checkout or
    /root/reference): the GPU box of the driver has none, and the field then says so."""
    import types
    root = os.environ.get("FX_REFERENCE_ROOT")
    if not root:
        for cand in (os.path.join(ROOT, ".ref_checkout"), "/root/reference"):
            if os.path.isdir(os.path.join(cand, "fuxictr")):
                root = cand
                break
    if not root or not os.path.isdir(os.path.join(root, "fuxictr")):
        return {"skipped": "no reference checkout on this box (the reference is a Python package and does not "
                           "travel); builder-side figures: profiles/r05_dropin_timing.txt (host-only 0.43 - 0.54 M "
                           "samples/s for RankDataLoader, 3.49 M end to end behind the native DeviceNpzDataLoader)"}
    try:
        for name in ["polars", "h5py", "keras_preprocessing", "keras_preprocessing.sequence"]:
            sys.modules.setdefault(name, types.ModuleType(name))
        sys.modules["keras_preprocessing.sequence"].pad_sequences = lambda *a, **k: None
        sys.dont_write_bytecode = True
        if root not in sys.path:
            sys.path.insert(0, root)
        from fuxictr.pytorch.dataloaders import RankDataLoader
        fmap, spec = synthetic.criteo_feature_map(embedding_dim=16)
        rng = np.random.default_rng(7)
        big = synthetic.criteo_batch(rng, args.batch * (batches + 6), dist="powerlaw")
        path = "/tmp/fx_ref_loader_rate_%d.npz" % os.getpid()
        np.savez(path, **{k: np.asarray(v) for k, v in big.items()})
        out = {"unit": "samples/sec", "by_num_workers": {},
               "what": "host side only: the reference's RankDataLoader (npz) for the c2 shape at batch %d on "
                       "%d host cores; a model behind it cannot be faster" % (args.batch, _host_cores())}
        for w in (0, 8):
            gen, _ = RankDataLoader(fmap, stage="train", train_data=path, batch_size=args.batch, shuffle=True,
                                    num_workers=w).make_iterator()
            it = iter(gen)
            for _ in range(3):
                next(it)
            t0 = time.perf_counter()
            n = 0
            for _ in range(batches):
                n += next(iter(next(it).values())).shape[0]
            out["by_num_workers"][str(w)] = n / (time.perf_counter() - t0)
            del it, gen
        os.remove(path)
        out["value"] = max(out["by_num_workers"].values())
        return out
    except Exception as exc:   # noqa: BLE001 - a reported sub-field must not end the run
        return {"skipped": "reference DataLoader failed here: %s: %s" % (type(exc).__name__, exc)}


def parity_full_vocab(case, gpu_index, steps=6, B=4096, dist="powerlaw"):
    """the native step against the oracle AT THE FULL VOCABULARY the metric is quoted on
    (33 762 603 rows; the parity tests of tests/baseline_shapes.py scale the tables x 0.01).  The native
    model's own initial state_dict is copied to the host and handed to the oracle (the reference's
    dense-gradient, dense-Adam algorithm on ATen's CPU kernels) and to a second oracle on ATen's GPU kernels
    (the yardstick: how far the reference's own two back ends drift apart); `steps` teacher-labelled batches go
    through all three — the native model through its real train_step (3 eager steps, the hipGraph capture,
    replays).  Batch 1 carries ids at the very top of every table (rows V-1, V-2, ...: the packed-row offsets
    and 32-bit keys of the 5-10 M row columns C3 / C12 / C16 / C21).
    Same weights -> logits within 1e-4 (before training, and at the oracle's trained weights loaded back
    into the native model); independent training -> per-step loss against the oracle, with the yardstick's
    own difference beside it.  The oracle is the CHECKER here, never the thing measured."""
    import tempfile
    from fuxictr_amd import zoo
    from oracle import ctr_oracle as O
    from tests import baseline_shapes as bs
    t_begin = time.perf_counter()
    model, features, cfg, spec, cards = bs.build(case, zoo, gpu_index, tempfile.mkdtemp(prefix="fx_parity_"),
                                                 vocab_scale=1.0, hip_graph=True)
    dev = model.device
    state0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    tr = O.OracleTrainer(cfg, state0, features, lr=1e-3, max_norm=10.0)
    yard = O.OracleTrainer(cfg, state0, features, lr=1e-3, max_norm=10.0, device=dev)
    rows = int(sum(cards) + len(cards))
    del state0
    teacher = bs.Teacher(features)
    rng = np.random.default_rng(2025)
    raw = bs.make_batches(case, spec, cards, rng, B, steps + 1, dist, teacher)
    # ids at the top of every table in batch 1 (ids are 1-based with 0 = padding: the largest id is V - 1)
    top = {}
    for name, fs in features.items():
        if fs["type"] == "categorical":
            V = int(fs["vocab_size"])
            n = min(256, V - 1)
            raw[1][name] = raw[1][name].copy()
            raw[1][name][:n] = (V - 1 - np.arange(n)).astype(raw[1][name].dtype)
            top[name] = V - 1
    hold = raw[steps]
    raw = raw[:steps]
    res = {"rows": rows, "steps": steps, "batch": B, "id_distribution": dist,
           "largest_table_rows": int(max(cards)), "top_ids_in_batch_1": True}
    torch.set_num_threads(min(_host_cores(), 32))

    def native_logits(b):
        return bs.logits_of(model, {k: v.to(dev) for k, v in bs.tb(b).items()})[0]
    model.eval()
    d0 = max(float(np.abs(native_logits(b) - tr.logits(bs.tb(b)).numpy()).max()) for b in (raw[0], raw[1]))
    res["max_dlogit_before"] = d0
    model.train()
    model._max_gradient_norm = 10.0
    ln, lo, ly = [], [], []
    for b in raw:
        t = bs.tb(b)
        ln.append(float(model.train_step({k: v.to(dev) for k, v in t.items()}).item()))
        lo.append(tr.train_step(t, t["label"])[0])
        ly.append(yard.train_step(t, t["label"])[0])
    model.optimizer.check_errors()
    res["loss_native"], res["loss_oracle"] = ln, lo
    res["max_dloss"] = float(np.abs(np.asarray(ln) - np.asarray(lo)).max())
    res["max_dloss_yardstick"] = float(np.abs(np.asarray(ly) - np.asarray(lo)).max())
    res["launch"] = "3 eager steps, hipGraph capture, %d replays" % max(steps - 4, 0)
    # independent trajectories: hold-out logits (how far apart two fp32 evaluations of the reference's
    # algorithm are after `steps` Adam steps is the yardstick's column)
    model.eval()
    ref = tr.logits(bs.tb(hold)).numpy()
    res["independent_training"] = {
        "mean_dlogit_native": float(np.abs(native_logits(hold) - ref).mean()),
        "mean_dlogit_yardstick": float(np.abs(yard.logits(bs.tb(hold)).numpy() - ref).mean())}
    # the oracle's TRAINED weights loaded into the native model (reference checkpoint keys): forward parity
    # at weights that have been through Adam, including the touched top rows
    del yard
    model.load_state_dict({k: v.detach().cpu() for k, v in tr.state.items()})
    model.eval()
    d1 = max(float(np.abs(native_logits(b) - tr.logits(bs.tb(b)).numpy()).max()) for b in (hold, raw[1]))
    res["max_dlogit_after"] = d1
    res["max_dlogit"] = max(d0, d1)
    res["seconds"] = round(time.perf_counter() - t_begin, 1)
    del model, tr
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return res


def parity_long_horizon(case, gpu_index, steps=400, B=4096, dist="powerlaw", vocab_scale=0.01,
                        holdout=131072, checkpoints=(100, 200, 400)):
    """a LONG native-vs-oracle trajectory.  The headline is timed on a state hundreds of
    steps old (rows 1 ... 300 steps behind the optimizer), while no comparison with the oracle was longer than
    12 steps.  Here the BASELINE shapes (tables x `vocab_scale`, so that the oracle's dense Adam over every
    row stays affordable) train for `steps` Adam steps on power-law ids — cold rows come back after hundreds of
    steps — through the native train_step (eager steps, hipGraph capture, replays) and through the oracle
    (rank_model.py:307-345 / torch_utils.py:72-76 restated on ATen's CPU kernels), with the oracle's identical
    code on ATen's GPU kernels as the same-run yardstick of how far two fp32 evaluations of the reference's
    own algorithm drift apart.  At every checkpoint the oracle's weights are loaded into a second native
    model (reference checkpoint keys): SAME WEIGHTS -> logits within 1e-4 and AUC / logloss within 5e-5 on
    `holdout` teacher-labelled rows.  The oracle is the CHECKER here, never the thing measured."""
    import tempfile
    from sklearn.metrics import log_loss, roc_auc_score
    from fuxictr_amd import zoo
    from oracle import ctr_oracle as O
    from tests import baseline_shapes as bs
    t_begin = time.perf_counter()
    root = tempfile.mkdtemp(prefix="fx_long_")
    model, features, cfg, spec, cards = bs.build(case, zoo, gpu_index, root, vocab_scale=vocab_scale,
                                                 hip_graph=True)
    probe = bs.build(case, zoo, gpu_index, root + "_probe", vocab_scale=vocab_scale)[0]
    dev = model.device
    state0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    tr = O.OracleTrainer(cfg, state0, features, lr=1e-3, max_norm=10.0)
    yard = O.OracleTrainer(cfg, state0, features, lr=1e-3, max_norm=10.0, device=dev)
    teacher = bs.Teacher(features)
    rng = np.random.default_rng(606)
    hold = bs.make_batches(case, spec, cards, rng, B, max(1, holdout // B), dist, teacher)
    y = np.concatenate([b["label"] for b in hold]).astype(np.float64)
    torch.set_num_threads(min(_host_cores(), 32))

    def to_dev(b):
        return {k: v.to(dev) for k, v in bs.tb(b).items()}

    def describe(lg, ref=None):
        pr = 1.0 / (1.0 + np.exp(-lg.astype(np.float64)))
        d = {"auc": float(roc_auc_score(y, pr)), "logloss": float(log_loss(y, pr))}
        if ref is not None:
            d["max_dlogit"] = float(np.abs(lg - ref).max())
            d["mean_dlogit"] = float(np.abs(lg - ref).mean())
        return d
    res = {"case": case, "rows": int(sum(cards) + len(cards)) if cards else None, "steps": steps, "batch": B,
           "id_distribution": dist, "vocab_scale": vocab_scale, "holdout_rows": int(len(y)),
           "launch": "3 eager steps, hipGraph capture, %d replays" % max(steps - 4, 0), "checkpoints": {}}
    model.train()
    model._max_gradient_norm = 10.0
    ln, lo, ly = [], [], []
    gaps = []
    for t in range(1, steps + 1):
        b = bs.make_batches(case, spec, cards, rng, B, 1, dist, teacher)[0]
        tb_ = bs.tb(b)
        ln.append(float(model.train_step(to_dev(b)).item()))
        lo.append(tr.train_step(tb_, tb_["label"])[0])
        ly.append(yard.train_step(tb_, tb_["label"])[0])
        if t in checkpoints:
            # how far behind the optimizer the rows of this state are (the exact mode's row stamps)
            for grp in model.optimizer._groups:
                if grp.last_step is not None and grp.D > 1:
                    ls = grp.last_step
                    touched = ls[ls > 0]
                    gaps.append({"step": t, "rows_touched": int(touched.numel()),
                                 "behind_max": int(t - touched.min().item()),
                                 "behind_mean": float((t - touched.float()).mean().item())})
                    break
            ref = np.concatenate([tr.logits(bs.tb(hb)).numpy() for hb in hold])
            probe.load_state_dict({k: v.detach().cpu() for k, v in tr.state.items()})
            probe.eval()
            same = describe(np.concatenate([bs.logits_of(probe, to_dev(hb))[0] for hb in hold]), ref)
            rf = describe(ref)
            ck = {"reference": rf,
                  "same_weights": {"max_dlogit": same["max_dlogit"], "dauc": abs(same["auc"] - rf["auc"]),
                                   "dlogloss": abs(same["logloss"] - rf["logloss"])}}
            if t == max(checkpoints):
                # the independently trained native model is only evaluated at the END: an evaluation flushes
                # the exact mode (every row caught up), and the point of the run is rows that are far behind
                model.eval()
                nat = describe(np.concatenate([bs.logits_of(model, to_dev(hb))[0] for hb in hold]), ref)
                model.train()
                yd = describe(np.concatenate([yard.logits(bs.tb(hb)).numpy() for hb in hold]), ref)
                ck["independent_native"] = {"mean_dlogit": nat["mean_dlogit"], "dauc": abs(nat["auc"] - rf["auc"]),
                                            "dlogloss": abs(nat["logloss"] - rf["logloss"])}
                ck["independent_yardstick"] = {"mean_dlogit": yd["mean_dlogit"], "dauc": abs(yd["auc"] - rf["auc"]),
                                               "dlogloss": abs(yd["logloss"] - rf["logloss"])}
            res["checkpoints"][str(t)] = ck
    model.optimizer.check_errors()
    ln, lo, ly = np.asarray(ln), np.asarray(lo), np.asarray(ly)
    res["row_age"] = gaps
    res["loss_first_last"] = [float(lo[0]), float(lo[-1])]
    res["max_dloss"] = float(np.abs(ln - lo).max())
    res["max_dloss_yardstick"] = float(np.abs(ly - lo).max())
    # the per-step differences by century (the trajectories of an ill-conditioned optimizer part over time)
    res["dloss_by_100"] = [[float(np.abs(ln - lo)[i:i + 100].max()), float(np.abs(ly - lo)[i:i + 100].max())]
                           for i in range(0, steps, 100)]
    res["seconds"] = round(time.perf_counter() - t_begin, 1)
    del model, probe, tr, yard
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return res


def _spawn_ranks(args):
    """`bench.py --gpus N` launched plainly: start N ranks of this script with torch.distributed.run
    (what the driver does for N > 1) and pass its exit code on."""
    import socket
    import subprocess
    if torch.cuda.device_count() < args.gpus and os.environ.get("FX_BENCH_BACKEND") != "gloo":
        sys.exit("bench.py --gpus %d: only %d HIP device(s) visible" % (args.gpus,
                                                                         torch.cuda.device_count()))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.exit(subprocess.call(cmd, env=env))


def make_pool(args, rank, cards, spec, dev, n_pool):
    """Synthetic batches resident in HBM (ids int64 / dense fp32 / label fp32 — what the reference's
    get_inputs would hold after .to(device)); distinct per rank and per step.  n_pool distinct
    batches are drawn so that the rows a timed run touches (table + Adam state, ~25 K unique rows x
    408 B per batch) do NOT stay resident in the 256 MB Infinity Cache: every step reads rows it has
    not seen recently, as a real stream does."""
    from fuxictr_amd import synthetic
    rng = np.random.default_rng(1000 + rank)
    pool = []
    for _ in range(n_pool):
        if args.model == "DIN":
            b = synthetic.taobao_batch(rng, args.batch, spec, dist=args.dist)
        else:
            b = synthetic.criteo_batch(rng, args.batch, cards=cards, dist=args.dist)
        if args.zoo == "reference" and args.model == "DLRM":
            # the reference's DLRM.forward concatenates the numeric columns on dim -1 (DLRM.py:115): they
            # must arrive as [B, 1], which is how its own data loader shapes them
            b = {k: (v.reshape(-1, 1) if (v.dtype == np.float32 and k != "label") else v)
                 for k, v in b.items()}
        if args.host_inputs:
            # what the reference's DataLoader yields: host tensors, int64 ids / float64 numerics
            pool.append({k: torch.from_numpy(v.astype(np.float64) if v.dtype == np.float32 else v)
                         for k, v in b.items()})
        else:
            pool.append({k: torch.from_numpy(v).to(dev) for k, v in b.items()})
    return pool


def probe_losses(args, model, cards, spec, rank, world, dev, dist, sharded):
    """--probe-loss: the same model and the same global batch on any number of ranks -> the same losses."""
    import zlib
    from fuxictr_amd import synthetic
    G = args.batch * max(world, args.probe_world, 1)
    ref = model.full_state_dict() if sharded else model.state_dict()
    full = {}
    for k in sorted(ref):
        v = ref[k]
        if not torch.is_floating_point(v):
            continue
        g = torch.Generator().manual_seed(zlib.crc32(k.encode()) & 0x7FFFFFFF)
        scale = 0.0 if k.endswith(".bias") else (0.01 if ".embedding_layers." in k else 0.03)
        full[k] = (torch.randn(v.shape, generator=g) * scale).to(torch.float32)
    if sharded:
        model.load_full_state_dict({k: v.to(dev) for k, v in full.items()})
    else:
        model.load_state_dict({k: v.to(dev) for k, v in full.items()}, strict=False)
    rng = np.random.default_rng(777)
    use_graph, model._use_graph = model._use_graph, False
    lo, n_loc = rank * G // max(world, 1), G // max(world, 1)
    out = []
    from fuxictr_amd import layers as _layers
    _layers.A2A_FILL_PROBE.update(on=sharded, max_used=0)
    for _ in range(2):
        b = synthetic.taobao_batch(rng, G, spec, dist=args.dist) if args.model == "DIN" \
            else synthetic.criteo_batch(rng, G, cards=cards, dist=args.dist)
        mine = {k: torch.from_numpy(np.ascontiguousarray(v[lo:lo + n_loc])).to(dev) for k, v in b.items()}
        if use_graph:
            # eager, but on the stream the later capture uses (autograd's AccumulateGrad nodes are tied to
            # the stream of the first backward: a probe on the default stream broke the segmented capture)
            if not getattr(model.optimizer, "_max_norm_explicit", False):
                model.optimizer.set_max_norm(model._max_gradient_norm, _from_model=True)
            model.optimizer.sync_lr()
            loss = model._side_stream_step(mine)
        else:
            loss = model.train_step(mine)
        loss = loss.detach().reshape(1).to(torch.float64)
        if dist is not None:
            dist.all_reduce(loss, op=dist.ReduceOp.SUM)
            loss = loss / world
        out.append(float(loss.item()))
    model._use_graph = use_graph
    _layers.A2A_FILL_PROBE["on"] = False
    if sharded and _layers.A2A_FILL_PROBE["cap"]:
        used = torch.tensor([float(_layers.A2A_FILL_PROBE["max_used"])], dtype=torch.float64, device=dev)
        if dist is not None:
            dist.all_reduce(used, op=dist.ReduceOp.MAX)
        PROBE_EXTRA["a2a_bucket"] = {
            "fullest_bucket_rows": int(used.item()), "capacity_rows": int(_layers.A2A_FILL_PROBE["cap"]),
            "even_share_of_lookups": int(_layers.A2A_FILL_PROBE["even_share"]),
            "fill": float(used.item()) / _layers.A2A_FILL_PROBE["cap"],
            "note": "unique keys one rank sends to one owner in the probe steps (max over ranks, owners, "
                    "steps) against the fixed per-owner capacity of the all-to-all blocks; FX_FLAG_A2A_OVERFLOW "
                    "would abort the run (optimizer.check_errors)"}
    return out


PROBE_EXTRA = {}
SPARSE_RECORD_STEPS = 32
C5_BATCH = 32768


def gather_at_c5_batch(args, model, cards, dev, n_batches=12):
    """The fused gather (k_emb_fm_fwd / k_emb_gather_fwd) at configs[4]'s batch of 32 768 on this GPU's
    full tables: the forward of `n_batches` distinct batches is recorded and the gather launches are
    replayed round-robin (12 x ~55 MB of looked-up rows, 82 MB record each).  At B = 4096 the kernel is
    latency-bound (one HBM latency of data in flight); this is the size at which it shows bandwidth
    (north_star: "achieved HBM-bandwidth fraction on the embedding gather")."""
    from fuxictr_amd import ops, synthetic
    rng = np.random.default_rng(4242)
    big = []
    for _ in range(n_batches):
        b = synthetic.criteo_batch(rng, C5_BATCH, cards=cards, dist=args.dist)
        big.append({k: torch.from_numpy(v).to(dev) for k, v in b.items()})
    was_training = model.training
    model.eval()
    ops.KernelTimer.reset()
    ops.KernelTimer.recording = True
    with torch.no_grad():
        for b in big:
            model.forward(b)
            ops.KernelTimer.next_step()
    ops.KernelTimer.recording = False
    torch.cuda.synchronize(dev)
    kt = ops.KernelTimer.replay(reps=10)
    ops.KernelTimer.reset()
    model.train(was_training)
    for key in ("k_emb_fm_fwd@alone", "k_emb_gather_fwd@alone"):
        if key in kt and kt[key]["total_ms"] > 0:
            out = dict(kt[key])
            out["kernel"] = key.split("@")[0]
            return out
    return None


def measure(args, rank, local_rank, world, world1, dev, dist):
    """Build args.model, warm up, time exactly args.steps steps, then the instrumented pass.
    -> dict(dt, ktimes, launch, parallelism, cards, timing_mode, rows)."""
    from fuxictr_amd import ops, synthetic
    cards = [max(3, int(c * args.vocab_scale)) for c in synthetic.CRITEO_CARDS]
    parallelism = "single GPU"
    if (world > 1 or world1) and not args.replicas:
        # row-sharded tables (row % world) + all-to-all exchange + dense all-reduce
        model, fmap, spec = build_model(args, local_rank, cards, shard="row")
        parallelism = ("tables row-sharded over %d ranks (RCCL all-to-all of ids / rows / row "
                       "gradients), towers data-parallel (one flat all-reduce carrying the clip "
                       "norm partials); " % world)
    else:
        model, fmap, spec = build_model(args, local_rank, cards)
        if world > 1:
            parallelism = "%d independent replicas (--replicas; no data-path collective)" % world
    model.train()
    probe = None
    if args.probe_loss:
        probe = probe_losses(args, model, cards, spec, rank, world, dev, dist,
                             sharded=(world > 1 or world1) and not args.replicas)
    # graph mode: `age` steps bring the run to the state a training run sustains, THEN --warmup more steps, then
    # the clock (round 6: the flag means what it says; `aged_steps` is reported beside it)
    age = max(args.age_steps, 20) if (not args.no_graph and not args.loader
                                      and not args.host_inputs) else max(args.warmup, 20)
    n_pool = min(args.pool, max(8, age + max(args.warmup, 0) + 3 * args.steps + 40))
    pool = make_pool(args, rank, cards, spec, dev, n_pool)

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

    def prepare_pool():
        # host side of the input cast of every pool batch (pointer blocks of the one pack launch),
        # once per batch object: what an epoch over HBM-resident batches pays on its first pass only
        if getattr(model, "_graph_state", None) is not None and loader_iter is None \
                and not args.host_inputs:
            for b in pool:
                model.prepare_batch(b)
    young = None
    aged = 0
    try:
        # graph mode: 3 eager steps + the capture, then the host side of the pool's input casts, then replays
        # back to back right up to the clock (the chip's power management needs ~20 ms of sustained load to
        # settle, profiles/r04_step_spread.txt).  The run is AGED before the clock: `age` steps on distinct
        # batches, so that the rows a timed step touches carry the revisit distances of a long run (the
        # exact-mode catch-up replays missed steps: k_catchup_rows ran 11.6 us at step 20 and 55 us at
        # step 300 in round 4).  The same number of steps timed right after step 20 is kept as `young_run`.
        warm_run = age if model._use_graph else max(args.warmup, 0)
        for i in range(min(20, warm_run)):
            if i == 4 and model._use_graph:
                prepare_pool()
            model.train_step(next_batch(step_i))
            step_i += 1
        if warm_run > 20 + args.steps:
            sync()
            y0 = torch.cuda.Event(enable_timing=True)
            y1 = torch.cuda.Event(enable_timing=True)
            y0.record()                      # (created here; recorded again below)
            y1.record()
            model.train_step(next_batch(step_i))     # the same bracket as the headline's: see below
            step_i += 1
            sync()
            y0.record()
            for _ in range(args.steps):
                model.train_step(next_batch(step_i))
                step_i += 1
            y1.record()
            sync()
            young = {"ms_per_step": y0.elapsed_time(y1) / args.steps, "steps_before": 21,
                     "steps": args.steps}
        while step_i < warm_run:
            model.train_step(next_batch(step_i))
            step_i += 1
        aged = step_i if model._use_graph else 0
        if model._use_graph:
            for _ in range(max(args.warmup, 0)):
                model.train_step(next_batch(step_i))
                step_i += 1
        warm_run = step_i
        sync()
    except Exception as exc:   # noqa: BLE001 — e.g. a capture problem on a software stack not seen
        if not model._use_graph:                      # in development: keep the run, launch eagerly
            raise
        launch_note = "hipGraph capture failed (%s: %s); eager launches" % (type(exc).__name__, exc)
        print("[bench] " + launch_note, file=sys.stderr, flush=True)
        model._use_graph = False
        model._graph_state = None
        warm_run = args.warmup
        for _ in range(args.warmup):
            model.train_step(pool[step_i % n_pool])
            step_i += 1
        sync()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    # (the HIP events behind torch's Event objects are created at their first record: done before the clock)
    ev0.record()
    ev1.record()
    # One more UNTIMED step between two synchronises: the first launch after a LONG blocking synchronise (the one
    # that drained the warm phase) costs the launching thread 0.1 - 1.2 ms (measured: 1.16 ms for the first
    # train_step of the DIN leg, 39 us with this step in front; profiles/r06_soak_and_step_count.txt) — host wake-up,
    # not GPU work, and it used to sit inside a 9 - 14 ms clock.  The clock still starts right after a barrier +
    # torch.cuda.synchronize(); the step is counted in `steps_before_clock`.
    model.train_step(next_batch(step_i))
    step_i += 1
    warm_run += 1
    sync()
    t0 = time.perf_counter()
    ev0.record()                 # on the launch stream (torch's current stream): first launch ...
    for _ in range(args.steps):
        model.train_step(next_batch(step_i))
        step_i += 1
    ev1.record()                 # ... to the end of the last kernel, without the host's sync latency
    sync()
    dt = time.perf_counter() - t0
    dt_events = 1e-3 * ev0.elapsed_time(ev1)
    model.optimizer.check_errors()
    # per-step spread: a SECOND pass of the same number of steps with one event per step (the events
    # cost a few us of launch boundary each, so this pass is not the headline; it shows whether the
    # mean hides a ramp or outliers)
    step_us = None
    if not args.no_step_events:
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        sync()
        evs[0].record()
        for j in range(args.steps):
            model.train_step(next_batch(step_i))
            step_i += 1
            evs[j + 1].record()
        sync()
        us = sorted(1e3 * evs[j].elapsed_time(evs[j + 1]) for j in range(args.steps))
        step_us = {"min": us[0], "median": us[len(us) // 2], "p90": us[min(len(us) - 1, int(0.9 * len(us)))],
                   "max": us[-1], "mean": sum(us) / len(us),
                   "first3": [1e3 * evs[j].elapsed_time(evs[j + 1]) for j in range(min(3, args.steps))],
                   "note": "second pass of %d steps, one HIP event per step on the launch stream "
                           "(events add launch boundaries: not the headline)" % args.steps}
    timing_mode, ktimes = None, {}
    if not args.no_kernel_timing:
        # Per-kernel durations cannot be observed inside a replayed hipGraph, and an event pair around
        # a single launch adds ~10 us on this platform.  So ONE eager step over the next batch of the
        # pool is RECORDED (every native entry point with its live arguments) and each recorded launch
        # group (one GEMM shape, the sparse path) is captured in step order into one hipGraph that is
        # replayed 20x between one event pair (ops.KernelTimer): the average is the kernel's duration
        # plus the ~1.3 us dependent-launch boundary, i.e. what rocprofv3's kernel trace of the timed
        # region reports (profiles/).
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        use_graph, model._use_graph = model._use_graph, False
        for _ in range(2):                             # eager warm-up of the non-graph path
            model.train_step(pool[step_i % n_pool])
            step_i += 1
        sync()
        ops.KernelTimer.reset()
        ops.KernelTimer.recording = True
        n_rec = min(SPARSE_RECORD_STEPS, n_pool)
        for j in range(n_rec):
            # one eager step per DISTINCT batch; the HBM-bound groups are replayed round-robin over all
            # of them (32 x ~10 MB of rows + Adam state > the 256 MB Infinity Cache, as in the timed
            # region), the GEMM groups over the first one
            model.train_step(pool[step_i % n_pool])
            step_i += 1
            ops.KernelTimer.next_step()
        ops.KernelTimer.recording = False
        sync()
        ktimes = ops.KernelTimer.replay(reps=20)
        ops.KernelTimer.reset()
        if (args.model != "DIN" and world == 1 and not world1 and not args.host_inputs
                and not args.loader):
            ktimes["gather_b32768"] = gather_at_c5_batch(args, model, cards, dev)
        model._use_graph = use_graph
        timing_mode = ("every native launch of %d eager steps over %d distinct batches recorded; the "
                       "launches of one GEMM shape (first recorded step) / of the sparse path (ALL "
                       "recorded steps, round-robin, so every launch meets rows it has not seen for %d "
                       "batches - more than the 256 MB Infinity Cache holds) are captured in step order "
                       "into a hipGraph that is replayed 20x between one HIP-event pair on the launch "
                       "stream (the timed region itself replays a hipGraph); average = kernel + "
                       "dependent-launch boundary; a split-K GEMM call includes its k_splitk_reduce "
                       "launch, so the GEMM figure is a few per cent below the k_gemm_f32_* rows of the "
                       "rocprofv3 summary under profiles/" % (n_rec, n_rec, n_rec - 1))
    if dist is not None:
        t = torch.tensor([dt, dt_events], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, dt_events = float(t[0].item()), float(t[1].item())
    launch = launch_note or ("hipGraph replay" if model._use_graph else "eager")
    if parallelism.endswith("; "):
        # which of the two captured forms of the sharded step actually ran (fuxictr_amd/dist.py)
        parallelism += ("eager launches" if not model._use_graph else
                        (getattr(model._dist, "graph_mode", None) or "hipGraph"))
    if model._dist is not None:
        model.release_graphs()       # recorded RCCL kernels must be gone before the communicator
    rows = sum(cards) + len(cards)
    del model, pool
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return {"dt": dt, "dt_events": dt_events, "step_us": step_us, "warmup_run": warm_run, "young": young,
            "aged": aged,
            "ktimes": ktimes, "launch": launch, "parallelism": parallelism,
            "cards": cards, "timing_mode": timing_mode, "rows": rows, "n_pool": n_pool,
            "probe": probe}


def _traffic(model, batch, world):
    """Fabric-side bytes per GEMM launch from the committed PMC passes (rocprofv3 --pmc cannot run
    inside this process); only valid for the exact workload it was collected on."""
    if model != "DeepFM" or batch != 4096 or world != 1:
        return None, None
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json"):
        try:
            with open(os.path.join(here, name)) as f:
                return json.load(f)["traffic_bytes_per_launch"], name
        except (OSError, ValueError, KeyError):
            continue
    return None, None


SPARSE_LAUNCH_NAMES = ("dedup_catchup", "k_emb_fm_fwd", "emb_fm_bwd", "sparse_update_multi", "sparse_adam",
                       "adam_catchup", "adam_catchup_rows", "sparse_sgd", "lr_fwd")


def _sparse_traffic(model, batch, world, kernel=None):
    """Fabric-side bytes of the sparse-path kernels from the committed PMC passes (per step, or of one
    kernel per launch); only valid for the exact workload it was collected on."""
    if model != "DeepFM" or batch != 4096 or world != 1:
        return None
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json"):
        try:
            with open(os.path.join(here, name)) as f:
                d = json.load(f)
            unit = ("bytes (L2 fabric requests incl. Infinity-Cache hits, FETCH x2 + WRITE — the x2 is the "
                    "gfx950 correction for wide coalesced reads and overstates 64-byte row reads; "
                    "profiles/%s)" % name)
            if kernel is None:
                return d["sparse_traffic_bytes_per_step"], unit + " per step"
            pk = d["per_kernel_bytes_per_launch"]
            # (round 4's gather kernel is k_emb_fm_fwd2; a file that only knows the first version's name
            # does not describe it)
            if kernel + "2" in pk:
                return pk[kernel + "2"], unit + " per launch"
            if name.startswith(("r04", "r05", "r06")):
                return pk[kernel], unit + " per launch"
            return None
        except (OSError, ValueError, KeyError):
            continue
    return None


# fp32 record row written by the gather launch for the tower's first GEMM: F * D * 4 bytes
RECORD_BYTES_PER_SAMPLE = {"DeepFM": 39 * 16 * 4, "DCNv2": 39 * 16 * 4, "xDeepFM": 39 * 16 * 4}


def _gather_bytes_per_sample(args):
    # Fs (4 D + 4) + Fd 4 (+ Fs 4 for the D = 1 first-order rows of DeepFM / xDeepFM)
    return 1924.0 if args.model in ("DeepFM", "xDeepFM") else 1820.0


def rooflines(m, args, world):
    """roofline objects from the instrumented pass of one measured workload."""
    out = {}
    kt = m["ktimes"]
    n_inst = 1                         # the record is ONE step
    # the MFMA kernel's launches: every GEMM shape with all three extents >= 64 (the 1-wide head
    # of the tower and its two gradients run on the skinny HBM-bound kernels, not on k_gemm_f32_pipe)
    # ("gemm MxNxK": one product; "gemm2 MxNxK": the dW + dX pair of a layer as one launch, 4 M N K flops)
    def dims(k):
        return [int(x) for part in k.split(" ", 1)[1].split("+") for x in part.split("x")]
    # ("gemmN a+b+...": several independent products of one depth as ONE grid — DCNv2's cross and deep
    # layer, forward, or their dW / dX products, backward)
    mf = [v for k, v in kt.items() if k.startswith(("gemm ", "gemm2 ", "gemmN ")) and min(dims(k)) >= 64]
    g = None
    if mf:
        g = {"launches": sum(v["launches"] for v in mf), "total_ms": sum(v["total_ms"] for v in mf),
             "work": sum(v["work"] for v in mf)}
        g["avg_us"] = 1e3 * g["total_ms"] / max(g["launches"], 1)
    if g and g["total_ms"] > 0:
        ach = g["work"] / (g["total_ms"] * 1e-3) / 1e12
        traffic, src = _traffic(args.model, args.batch, world)
        out["roofline"] = {"kernel": ("k_gemm_x6 / k_gemm_x6_multi (split-bf16 GEMM: operands split into three "
                                      "exact bf16 planes in the kernel, six v_mfma_f32_32x32x16_bf16 products "
                                      "per fp32 product, fp32 accumulate; MLP / CrossNet forward, dW + dX "
                                      "grids backward)") if X6_ON else
                                     "k_gemm_f32_pipe / k_gemm_f32_pair (fp32 MFMA GEMM, MLP/CrossNet "
                                     "fwd; dW+dX pairs bwd)",
                           "bound": "mfma", "achieved": ach, "peak": PEAK_GEMM_TFLOPS,
                           "unit": "TFLOP/s", "frac": ach / PEAK_GEMM_TFLOPS,
                           "peak_note": ("algorithmic (fp32) flops 2MNK per product; the kernel executes 6 bf16 "
                                         "MFMA flops per algorithmic flop, so its matrix-core peak is the dense "
                                         "bf16 peak / 6 = %.1f TFLOP/s; against the fp32-MFMA peak (%.1f) the "
                                         "same figure is %.3f" % (PEAK_GEMM_TFLOPS, PEAK_FP32_MFMA_TFLOPS,
                                                                  ach / PEAK_FP32_MFMA_TFLOPS)) if X6_ON else None,
                           "traffic": traffic,
                           "traffic_unit": "bytes per launch (L2 fabric requests incl. Infinity-"
                                           "Cache hits; profiles/%s)" % src if src else None,
                           "launches": g["launches"], "avg_launch_us": g["avg_us"],
                           "gemm_us_per_step": 1e3 * g["total_ms"] / n_inst,
                           "timing": m["timing_mode"]}
        shapes = {}
        for k, v in kt.items():
            if k.startswith(("gemm ", "gemm2 ", "gemmN ")) and v["total_ms"] > 0:
                tf = v["work"] / (v["total_ms"] * 1e-3) / 1e12
                label = k.split(" ", 1)[1] + (" (dW+dX pair, incl. slab reduce)"
                                              if k.startswith("gemm2 ") else
                                              " (one grid, incl. slab reduces)"
                                              if k.startswith("gemmN ") else "")
                shapes[label] = {"launches_per_step": v["launches"] / n_inst,
                                 "avg_launch_us": round(v["avg_us"], 2),
                                 "tflops": round(tf, 1),
                                 "frac": round(tf / PEAK_GEMM_TFLOPS, 3)}
        out["roofline"]["by_shape_MxNxK"] = shapes
    sp = kt.get("sparse_path")
    if sp and sp["total_ms"] > 0:
        # train-step upper bound (no duplicates) per sample — 12 532 B DeepFM/xDeepFM
        # (gather 1924 + 26 x (384 + 24) sparse Adam), 11 804 B models without the LR copy
        per_sample = 12532 if args.model in ("DeepFM", "xDeepFM") else 11804
        if args.model == "DIN":
            per_sample = 4352 + 64 * 384
        us_step = 1e3 * sp["total_ms"] / n_inst
        ach = per_sample * args.batch / (us_step * 1e-6) / 1e9
        out["roofline_sparse"] = {
            "kernels": "de-dup + catch-up + gather(+LR/FM) + gradient run-reduce + sparse-row Adam "
                       "(every launch of the embedding path of one step)",
            "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": ach / PEAK_HBM_GBS, "traffic": None,
            "algorithmic_bytes_per_sample": per_sample, "us_per_step": us_step,
            "launches_per_step": sp["launches"] / n_inst,
            "distinct_batches_replayed": sp.get("steps", 1),
            "per_kernel": "profiles/r06_step_timeline_%s_final.txt (rocprofv3 kernel trace of one "
                          "step of the timed region)" % args.model}
        tr = _sparse_traffic(args.model, args.batch, world)
        if tr is not None:
            out["roofline_sparse"]["traffic"] = tr[0]
            out["roofline_sparse"]["traffic_unit"] = tr[1]
    da = kt.get("din_attention")
    if da and da["total_ms"] > 0:
        # the fused DIN attention passes (fx_din_attn.hip).  Algorithmic flops = what the reference's
        # graph needs: the hidden-layer product 2 (B L) 4E H once forward, twice backward (dW1, dX) —
        # the statistics passes and the recomputation the fused form adds are NOT counted as work
        ach = da["work"] / (da["total_ms"] * 1e-3) / 1e12
        out["roofline_attention"] = {
            "kernels": "k_din_attn(2)_{stats,fwd,bwd_sums,bwd} + their partial-sum launches",
            "bound": "mfma", "achieved": ach, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": ach / PEAK_FP32_MFMA_TFLOPS, "traffic": None,
            "us_per_step": 1e3 * da["total_ms"] / n_inst, "launches_per_step": da["launches"] / n_inst,
            "note": "serial chain per wave (x tile -> MFMA -> Dice gate -> ...), 4 passes around the "
                    "batch statistics; see DESIGN.md section 4"}
    cn = kt.get("cin")
    if cn and cn["total_ms"] > 0:
        # the CIN passes (fx_cin_mfma.hip: forward, dX, dW of every layer).  Algorithmic flops = the
        # compress product 2 O F0 Mi D per sample, once forward and twice backward — the padding of the
        # MFMA tiles (Mi 39 -> 40 / 48) and the tile arithmetic on the VALU are not counted as work
        ach = cn["work"] / (cn["total_ms"] * 1e-3) / 1e12
        out["roofline_cin"] = {
            "kernels": "k_cin_{fwd,dx,dw}_mfma of the three CIN layers (v_mfma_f32_16x16x4_f32)",
            "bound": "mfma", "achieved": ach, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": ach / PEAK_FP32_MFMA_TFLOPS, "traffic": None,
            "us_per_step": 1e3 * cn["total_ms"] / n_inst, "launches_per_step": cn["launches"] / n_inst}
    e, ename = kt.get("k_emb_fm_fwd@alone"), "k_emb_fm_fwd (gather + numeric expansion + LR + FM, one launch)"
    if not (e and e["total_ms"] > 0):
        e, ename = kt.get("k_emb_gather_fwd"), "k_emb_gather_fwd"
    if e and e["total_ms"] > 0:
        ach = e["work"] / (e["total_ms"] * 1e-3) / 1e9
        out["roofline_gather"] = {"kernel": ename, "bound": "hbm", "achieved": ach,
                                  "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                  "frac": ach / PEAK_HBM_GBS, "traffic": None,
                                  "launches": e["launches"], "avg_launch_us": e["avg_us"],
                                  "distinct_batches_replayed": e.get("steps", 1),
                                  "note": "B = %d: the recorded gather launches of %d distinct batches "
                                          "replayed round-robin (rows not cache-resident, as in the "
                                          "step); about one HBM latency of data in flight - latency-"
                                          "bound at this batch size" % (args.batch, e.get("steps", 1))}
        tr = _sparse_traffic(args.model, args.batch, world, kernel="k_emb_fm_fwd")
        if tr is not None:
            out["roofline_gather"]["traffic"] = tr[0]
            out["roofline_gather"]["traffic_unit"] = tr[1]
    gb = kt.get("gather_b32768")
    if gb and gb.get("total_ms", 0) > 0:
        ach = gb["work"] / (gb["total_ms"] * 1e-3) / 1e9
        out["roofline_gather_b32768"] = {
            "kernel": gb["kernel"] + " at configs[4]'s batch (32 768 samples) on this GPU's full tables",
            "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": ach / PEAK_HBM_GBS, "traffic": None, "launches": gb["launches"],
            "avg_launch_us": gb["avg_us"], "distinct_batches_replayed": gb.get("steps", 1),
            "algorithmic_bytes_per_launch": gb["work"] / max(gb["launches"], 1e-9)}
        # SURVEY 8d counts the [B, F, D] record write "only if the kernel is not fused into the consumer": the
        # interaction terms ARE computed in this launch, but the first GEMM still reads the record from HBM, so
        # the launch also writes F * D * 4 bytes per sample — the bytes it really moves are given beside the
        # algorithmic figure (round 6; the algorithmic fraction cannot pass ~0.43 while the record is written)
        n_b = gb["work"] / max(gb["launches"], 1e-9) / max(_gather_bytes_per_sample(args), 1e-9)
        rec = float(n_b) * RECORD_BYTES_PER_SAMPLE.get(args.model, 0)
        if rec > 0:
            g = out["roofline_gather_b32768"]
            moved = g["algorithmic_bytes_per_launch"] + rec
            g["with_record_write"] = {"bytes_per_launch": moved,
                                      "achieved": moved / (gb["avg_us"] * 1e-6) / 1e9,
                                      "frac": moved / (gb["avg_us"] * 1e-6) / 1e9 / PEAK_HBM_GBS,
                                      "note": "algorithmic reads + the [B, 39, 16] fp32 record this launch "
                                              "writes for the first GEMM"}
    return out


def timing_detail(m, args):
    """Where the wall clock of the timed region goes: the same region between one HIP-event pair on the
    launch stream, the per-step spread of a second pass, and the sum of the step's kernels (replayed
    per launch group by ops.KernelTimer; DeepFM / DCNv2, whose every entry point is recorded)."""
    out = {"ms_per_step_events": 1e3 * m["dt_events"] / args.steps,
           "warmup_steps_run": m["warmup_run"]}
    if m.get("young"):
        y = m["young"]
        out["young_run"] = {"ms_per_step": round(y["ms_per_step"], 4),
                            "value": args.batch * 1e3 / y["ms_per_step"],
                            "note": "%d steps timed right after step %d of the same run (HIP events), NOT the "
                                    "headline; `value` is measured after %d steps.  Until round 5 a young run "
                                    "was 4 - 6 %% faster (rows had few missed Adam steps to replay); with the Adam "
                                    "series table (round 6) the catch-up does not depend on a row's age"
                                    % (y["steps"], y["steps_before"], m["warmup_run"])}
    if m.get("step_us"):
        out["step_us"] = {k: (round(v, 1) if isinstance(v, float) else
                              [round(x, 1) for x in v] if isinstance(v, list) else v)
                          for k, v in m["step_us"].items()}
    tot = (m.get("ktimes") or {}).get("__step__")
    if tot and args.model in ("DeepFM", "DCNv2"):
        out["kernel_sum_us"] = round(1e3 * tot["total_ms"], 1)
        out["kernel_sum_launches"] = round(tot["launches"], 1)
        out["wall_minus_kernel_sum_us"] = round(1e3 * m["dt"] / args.steps * 1e3 - 1e3 * tot["total_ms"], 1)
    return out


def workload_name(args, rows):
    if args.model == "DIN":
        return ("configs[3]: DIN on synthetic Taobao-shape sequences (14 categorical + "
                "click_sequence len 50 sharing adgroup_id, emb_dim 16, attention [64] Dice, dnn "
                "[512,128,64]), Adam, full training step")
    return ("configs[%d]: %s on synthetic Criteo (26 sparse + 13 dense, %d rows, emb_dim 16, %s), "
            "Adam, full training step"
            % ({"DeepFM": 1, "DCNv2": 2, "DLRM": 4}.get(args.model, 1), args.model, rows,
               {"DCNv2": "MLP 4x1024, 3 cross layers",
                "DLRM": "bottom MLP [512,256,16], dot interaction, top MLP [1024,1024,512,256]",
                "xDeepFM": "MLP 4x1024, CIN [16,16,16]"}.get(args.model, "MLP 4x1024")))


_JSON_FD = None


def _own_stdout():
    """ONE JSON line on stdout, whatever the libraries print: RCCL ("RCCL version : ..." at the flush of
    its stdio buffer when the process ends), gloo ("[Gloo] Rank 0 is connected ...") and hipBLASLt-style
    banners write to fd 1 from C code.  The line goes out through a private duplicate of the original
    stdout; fd 1 itself is pointed at stderr for the rest of the process — on every rank (torchrun merges
    the ranks' stdout into the launcher's)."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def _emit(line):
    os.write(_JSON_FD, (line + "\n").encode())


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _spawn_ranks(args)                                  # does not return
    _own_stdout()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d — the line must report the GPUs it ran on"
                 % (args.gpus, world))
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

    m = measure(args, rank, local_rank, world, world1, dev, dist)
    second = None
    if (args.model == "DeepFM" and world == 1 and not world1 and not args.no_dcnv2
            and not args.loader and not args.host_inputs):
        # BASELINE.json's metric names both models: the c3 DCNv2 step (CrossNet 624x624 GEMMs) is
        # measured in the same run and reported as a sub-object of the same line
        import copy
        args2 = copy.copy(args)
        args2.model = "DCNv2"
        m2 = measure(args2, rank, local_rank, world, world1, dev, dist)
        second = (args2, m2)
    fourth = None
    if (args.model == "DeepFM" and world == 1 and not world1 and not args.no_din
            and not args.loader and not args.host_inputs and args.zoo == "native"):
        # configs[3]: DIN on the synthetic Taobao-shape sequence schema (seq_len 50), same protocol
        import copy
        args4 = copy.copy(args)
        args4.model = "DIN"
        fourth = (args4, measure(args4, rank, local_rank, world, world1, dev, dist))
    third = None
    if (args.model == "DeepFM" and world == 1 and not world1 and not args.no_uniform and args.dist == "powerlaw"
            and not args.loader and not args.host_inputs):
        import copy
        args3 = copy.copy(args)
        args3.dist, args3.no_kernel_timing, args3.no_step_events = "uniform", True, True
        third = measure(args3, rank, local_rank, world, world1, dev, dist)

    if rank == 0:
        global_batch = args.batch * world
        value = global_batch * args.steps / m["dt"]
        out = {
            "metric": "samples/sec at batch 4096, Criteo-shape DeepFM/DCNv2, 1/2/4/8 MI355X",
            "value": value, "unit": "samples/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup if m["aged"] else m["warmup_run"], "aged_steps": m["aged"],
            "steps_before_clock": m["warmup_run"],
            "ms_per_step": 1e3 * m["dt"] / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (bf16x6 split operands, fp32 accumulate)" if X6_ON else "f32",
            "data": "synthetic",
            "config": {"workload": workload_name(args, m["rows"]),
                       "global_batch": global_batch, "per_gpu_batch": args.batch,
                       "id_distribution": args.dist, "sparse_update": args.sparse_update,
                       "emb_dtype": args.emb_dtype + (" (tables stored bf16, all arithmetic and "
                                                      "optimizer state fp32; NOT the headline "
                                                      "configuration)" if args.emb_dtype != "fp32"
                                                      else ""),
                       "launch": m["launch"],
                       "model_classes": ("the reference's own model_zoo classes behind "
                                         "fuxictr_amd.patch.install()" if args.zoo == "reference" else
                                         "fuxictr_amd.zoo (mirrors of the reference's model_zoo classes)"),
                       "distinct_batches": m["n_pool"],
                       "inputs": ("host tensors per step (DataLoader-style; one pinned staging "
                                  "copy per dtype) - PCIe-inclusive, NOT the headline number")
                       if args.host_inputs else
                       ("DeviceNpzDataLoader over a 64-batch synthetic .npz, shuffled (host column "
                        "gather + pinned H2D on a copy stream, prefetched) - end-to-end, NOT the "
                        "headline number") if args.loader else
                       "resident in HBM (%d distinct batches cycled: the rows touched exceed the "
                       "256 MB Infinity Cache)" % m["n_pool"],
                       "parallelism": m["parallelism"]},
        }
        out.update(timing_detail(m, args))
        out.update(rooflines(m, args, world))
        if m.get("probe") is not None:
            out["probe_loss"] = m["probe"]
            out.update(PROBE_EXTRA)
        if second is not None:
            args2, m2 = second
            sub = {"workload": workload_name(args2, m2["rows"]),
                   "value": args.batch * args.steps / m2["dt"], "unit": "samples/sec",
                   "ms_per_step": 1e3 * m2["dt"] / args.steps, "launch": m2["launch"]}
            sub.update(timing_detail(m2, args2))
            sub.update(rooflines(m2, args2, world))
            out["dcnv2"] = sub
        if fourth is not None:
            args4, m4 = fourth
            sub = {"workload": workload_name(args4, m4["rows"]),
                   "value": args.batch * args.steps / m4["dt"], "unit": "samples/sec",
                   "ms_per_step": 1e3 * m4["dt"] / args.steps, "launch": m4["launch"]}
            sub.update(timing_detail(m4, args4))
            sub.update(rooflines(m4, args4, world))
            out["din"] = sub
        if third is not None:
            m3 = third
            out["value_uniform"] = args.batch * args.steps / m3["dt"]
            out["uniform"] = {"value": out["value_uniform"], "unit": "samples/sec",
                              "ms_per_step": 1e3 * m3["dt"] / args.steps,
                              "warmup_steps_run": m3["warmup_run"],
                              "note": "the same DeepFM step on UNIFORM ids (SURVEY 8d: both distributions): "
                                      "~106 K unique rows per batch instead of ~25 K"}
        if world == 1 and not args.no_parity and not world1 and args.model in ("DeepFM", "DCNv2") \
                and args.vocab_scale == 1.0 and args.zoo == "native":
            out["parity_full_vocab"] = {}
            for case in (("c2_deepfm", "c3_dcnv2") if args.model == "DeepFM" and not args.no_dcnv2 else
                         ("c2_deepfm",) if args.model == "DeepFM" else ("c3_dcnv2",)):
                out["parity_full_vocab"][case] = parity_full_vocab(case, local_rank)
            out["parity_full_vocab"]["max_dlogit"] = max(v["max_dlogit"] for v in out["parity_full_vocab"].values())
            if args.long_parity_steps > 0 and args.model == "DeepFM":
                #  the long trajectory: tables x 0.01 so that the oracle's dense Adam is affordable
                r = parity_long_horizon("c2_deepfm", local_rank, steps=args.long_parity_steps,
                                        checkpoints=tuple(sorted({max(1, args.long_parity_steps // 4),
                                                                  max(1, args.long_parity_steps // 2),
                                                                  args.long_parity_steps})))
                out["parity_full_vocab"]["steps_%d" % args.long_parity_steps] = r
        if world == 1 and not world1:
            out["reference_dataloader"] = reference_dataloader_rate(args)
        if world == 1 and not args.no_cpu_baseline and args.model in ("DeepFM", "DCNv2"):
            out["cpu_baseline"] = cpu_baseline(args, m["cards"], args.cpu_baseline_steps)
        assert out["n_gpus"] == args.gpus
        _emit(json.dumps(out))
    if dist is not None:
        dist.barrier()
        # the captured steps (and the RCCL kernels recorded in them) were released in measure(); the
        # helper destroys the process group and ends the process itself should that still not return
        # (round 3: destroy_process_group hung while a graph with recorded collectives was alive,
        # profiles/r03_graph_collectives_teardown.txt)
        sys.stdout.flush()
        sys.stderr.flush()
        from fuxictr_amd.dist import DistContext
        DistContext.shutdown(timeout_s=20.0)


if __name__ == "__main__":
    main()
