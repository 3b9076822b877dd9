#!/usr/bin/env python3
"""Multi-seed spread of the baseline-shape training-parity statistics, with A/B switches.

VERDICT r2 weak #1: on ONE seed the c5 DLRM drifted further from the oracle than the two yardsticks
(the oracle with fp64 gradients, the oracle on ATen's GPU kernels) did.  One seed cannot separate a
kernel that loses accuracy from the +-lr noise Adam makes out of gradient rounding, so this script
measures the RMS over >= 8 model seeds of

    loss-trajectory difference, mean |dlogit| on the 64 k hold-out, dAUC, dlogloss

for native, ref64 and refgpu (all against the fp32 CPU oracle of the same seed), and repeats the native
leg under kernel switches (A/B): each switch is an environment variable read once per process, so every
(seed, dist, variant) is its own process; the oracle side of a (seed, dist) is computed once and shared
through an .npz.  Processes run side by side (the GPU box has 256 host cores).

  stage "oracle":  parity_sweep.py --stage oracle  case dist seed out.npz
  stage "native":  parity_sweep.py --stage native  case dist seed ref.npz variant
  driver:          parity_sweep.py --case c5_dlrm --seeds 1 2 ... --variants default,dot_valu,... --out f.jsonl

TEST INFRASTRUCTURE (imports oracle/): never used by the product.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

VARIANTS = {
    "default": {},
    "dot_valu": {"FX_DOT_MFMA": "0"},          # DLRM pairwise dot on the VALU kernels of round 1
    "no_pad": {"FX_MLP_PAD": "0"},             # 367-wide input NOT padded to 368 (unpipelined GEMM)
    "no_pair": {"FX_GEMM_PAIR": "0"},          # dW and dX as two launches
    "no_pipe": {"FX_GEMM_PIPE": "0"},          # the unpipelined GEMM kernel everywhere
    "splitk1": {"FX_DW_SPLITK": "1"},          # weight gradients without split-K (one k-ordered chain)
    "lazy": {"FX_PROBE_SPARSE": "lazy"},       # (not parity: SparseAdam semantics; scale reference)
    "no_inplace": {"FX_DIN_INPLACE": "0", "FX_DLRM_INPLACE": "0"},   # the concatenating compositions of DIN / DLRM
    "x6_off": {"FX_GEMM_BF16X6": "0"},         # round 5: every GEMM on the fp32-MFMA kernels
    "quad_off": {"FX_CATCHUP_QUAD": "0"},      # round 5: the plain catch-up replays
    "series_off": {"FX_CATCHUP_SERIES": "0"},  # round 6: no Adam series table (step-by-step replay of every gap)
    "buckets_off": {"FX_DEDUP_BUCKETS": "0"},  # round 6: sequence schemas on the device-wide radix sort (ascending rows)
    "record_off": {"FX_ROW_RECORD": "0"},      # round 6: table / m / v / last_step as four packed arrays
}


def _setup(case, dist, seed, steps, holdout, B):
    import numpy as np
    import baseline_shapes as BS
    from fuxictr_amd import zoo
    extra = {}
    if os.environ.get("FX_PROBE_SPARSE"):
        extra["sparse_update"] = os.environ["FX_PROBE_SPARSE"]
    model, features, cfg, spec, cards = BS.build(case, zoo, 0, "/tmp/fx_probe_%d" % os.getpid(),
                                                 seed=seed, **extra)
    teacher = BS.Teacher(features)
    rng = np.random.default_rng({"powerlaw": 11, "uniform": 12}[dist] + 1000 * seed)
    train = BS.make_batches(case, spec, cards, rng, B, steps, dist, teacher)
    test = BS.make_batches(case, spec, cards, rng, B, max(1, holdout // B), dist, teacher)
    return BS, model, features, cfg, train, test


def _describe(lgs, ref_lg, y):
    import numpy as np
    from sklearn.metrics import log_loss, roc_auc_score
    d = np.abs(lgs - ref_lg)
    p = 1.0 / (1.0 + np.exp(-lgs.astype(np.float64)))
    return {"max": float(d.max()), "mean": float(d.mean()), "auc": float(roc_auc_score(y, p)),
            "logloss": float(log_loss(y, p))}


def stage_oracle(case, dist, seed, out, steps, holdout, B):
    import numpy as np
    import torch
    torch.set_num_threads(int(os.environ.get("FX_PROBE_THREADS", "8")))
    from oracle import ctr_oracle as O
    BS, model, features, cfg, train, test = _setup(case, dist, seed, steps, holdout, B)
    state0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    del model
    tr = O.OracleTrainer(cfg, state0, features, lr=1e-3, max_norm=10.0)
    yards = {"ref64": O.OracleTrainer64(cfg, state0, features, lr=1e-3, max_norm=10.0)}
    if torch.cuda.is_available():
        yards["refgpu"] = O.OracleTrainer(cfg, state0, features, lr=1e-3, max_norm=10.0,
                                          device="cuda:0")
    lo, ly = [], {k: [] for k in yards}
    for b in train:
        t = BS.tb(b)
        lo.append(tr.train_step(t, t["label"])[0])
        for k, y_ in yards.items():
            ly[k].append(y_.train_step(t, t["label"])[0])
    lo = np.asarray(lo, dtype=np.float64)
    y = np.concatenate([b["label"] for b in test]).astype(np.float64)
    ref_lg = np.concatenate([tr.logits(BS.tb(b)).numpy() for b in test])
    ref = _describe(ref_lg, ref_lg, y)
    rows = []
    for k, y_ in yards.items():
        d = _describe(np.concatenate([y_.logits(BS.tb(b)).numpy() for b in test]), ref_lg, y)
        rows.append({"case": case, "dist": dist, "seed": seed, "who": k,
                     "loss": float(np.abs(np.asarray(ly[k]) - lo).max()), "max": d["max"],
                     "mean": d["mean"], "dAUC": d["auc"] - ref["auc"],
                     "dLL": d["logloss"] - ref["logloss"], "ref_auc": ref["auc"]})
    np.savez(out, ref_lg=ref_lg, y=y, lo=lo)
    for r in rows:
        print("ROW " + json.dumps(r), flush=True)


def stage_native(case, dist, seed, refnpz, variant, steps, holdout, B):
    import numpy as np
    import torch
    torch.set_num_threads(4)
    BS, model, features, cfg, train, test = _setup(case, dist, seed, steps, holdout, B)
    z = np.load(refnpz)
    ref_lg, y, lo = z["ref_lg"], z["y"], z["lo"]
    model.train()
    model._max_gradient_norm = 10.0
    ln = [float(model.train_step(BS.tb(b)).item()) for b in train]
    model.eval()
    lg = np.concatenate([BS.logits_of(model, BS.tb(b))[0] for b in test])
    model.optimizer.check_errors()
    d = _describe(lg, ref_lg, y)
    ref = _describe(ref_lg, ref_lg, y)
    print("ROW " + json.dumps({"case": case, "dist": dist, "seed": seed, "who": "native:" + variant,
                               "loss": float(np.abs(np.asarray(ln) - lo).max()), "max": d["max"],
                               "mean": d["mean"], "dAUC": d["auc"] - ref["auc"],
                               "dLL": d["logloss"] - ref["logloss"], "ref_auc": ref["auc"]}),
          flush=True)


def _run_all(jobs, par, log):
    """jobs: list of (argv, env).  Runs `par` at a time; returns the ROW dicts they printed."""
    rows, running, todo = [], [], list(jobs)
    while todo or running:
        while todo and len(running) < par:
            argv, env = todo.pop(0)
            p = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv,
                                 stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                 env=dict(os.environ, **env))
            running.append((p, argv))
        for item in list(running):
            p, argv = item
            if p.poll() is None:
                continue
            running.remove(item)
            so, se = p.communicate()
            got = [json.loads(l[4:]) for l in so.splitlines() if l.startswith("ROW ")]
            if p.returncode != 0 or not got:
                log.write("FAILED %s rc=%s\n%s\n" % (" ".join(argv), p.returncode, se[-1500:]))
                log.flush()
            rows += got
        time.sleep(0.2)
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stage")
    ap.add_argument("rest", nargs="*")
    ap.add_argument("--case", default="c5_dlrm")
    ap.add_argument("--dists", default="powerlaw,uniform")
    ap.add_argument("--seeds", type=int, nargs="*", default=list(range(1, 9)))
    ap.add_argument("--variants", default="default,dot_valu,no_pad,no_pair")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--holdout", type=int, default=65536)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--par", type=int, default=16)
    ap.add_argument("--out", default="gpurun_out/parity_sweep.jsonl")
    a = ap.parse_args()
    if a.stage == "oracle":
        case, dist, seed, out = a.rest
        return stage_oracle(case, dist, int(seed), out, a.steps, a.holdout, a.batch)
    if a.stage == "native":
        case, dist, seed, refnpz, variant = a.rest
        return stage_native(case, dist, int(seed), refnpz, variant, a.steps, a.holdout, a.batch)
    import numpy as np
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    log = open(a.out + ".log", "w")
    dists = a.dists.split(",")
    common = ["--steps", str(a.steps), "--holdout", str(a.holdout), "--batch", str(a.batch)]
    tmp = "/tmp/fx_parity_%d" % os.getpid()
    os.makedirs(tmp, exist_ok=True)
    t0 = time.time()
    jobs = [(["--stage", "oracle"] + common + [a.case, d, str(s), "%s/%s_%s_%d.npz" % (tmp, a.case, d, s)], {})
            for d in dists for s in a.seeds]
    rows = _run_all(jobs, a.par, log)
    log.write("oracle stage %.0f s\n" % (time.time() - t0))
    t0 = time.time()
    jobs = []
    for v in a.variants.split(","):
        for d in dists:
            for s in a.seeds:
                jobs.append((["--stage", "native"] + common +
                             [a.case, d, str(s), "%s/%s_%s_%d.npz" % (tmp, a.case, d, s), v], VARIANTS[v]))
    rows += _run_all(jobs, a.par, log)
    log.write("native stage %.0f s\n" % (time.time() - t0))
    with open(a.out, "w") as f:
        for r in rows:
            f.write(json.dumps(r) + "\n")
    # RMS table
    keys = ("loss", "mean", "max", "dAUC", "dLL")
    with open(a.out.replace(".jsonl", "") + "_rms.txt", "w") as f:
        f.write("%s: RMS over seeds %s of the difference to the fp32 CPU oracle (10 Adam steps, 64 k hold-out)\n"
                % (a.case, a.seeds))
        f.write("%-10s %-18s %3s " % ("dist", "who", "n") + " ".join("%11s" % k for k in keys) + "\n")
        for d in dists:
            whos = sorted(set(r["who"] for r in rows if r["dist"] == d))
            for w in whos:
                sel = [r for r in rows if r["dist"] == d and r["who"] == w]
                f.write("%-10s %-18s %3d " % (d, w, len(sel)) + " ".join(
                    "%11.3e" % float(np.sqrt(np.mean([r[k] ** 2 for r in sel]))) for k in keys) + "\n")
    print(open(a.out.replace(".jsonl", "") + "_rms.txt").read())


if __name__ == "__main__":
    main()
