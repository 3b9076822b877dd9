"""Worker of the 2-process tests of the row-sharded path (tests/test_dist_gloo.py on CPU with
emulated kernels; tests/test_gpu_dist.py with the real kernels, both ranks on cuda:0, collectives
staged through gloo).  Each rank trains on ITS HALF of the golden batches; the result must equal
the single-process reference run on the full batches."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)



def _shutdown(model):
    """Captured steps go before the communicator (RCCL kernels recorded into a hipGraph keep it alive)."""
    from fuxictr_amd.dist import DistContext
    DistContext.shutdown([model])

def run_c5(rank, world, port, out_path):
    """configs[4] in miniature: the c5 DLRM (bottom [512,256,16], dot, top [1024,1024,512,256],
    26 tables with the Criteo-skewed split) row-sharded over `world` ranks, each rank on its slice of
    the GLOBAL batch, against the oracle's single-process run on the full batches (CPU emulation of
    the kernels; tables scaled down so that dense Adam is affordable)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import _cpu_emul
    _cpu_emul.install_plain()
    import baseline_shapes as BS
    from fuxictr_amd import zoo
    from oracle import ctr_oracle as O
    model, features, cfg, spec, cards = BS.build("c5_dlrm", zoo, -1, "/tmp/fx_dist_c5_%d" % rank,
                                                 vocab_scale=0.0005, shard="row")
    # identical on every rank (tables all-gathered); cloned: the dense entries are live views
    state0 = {k: v.detach().clone() for k, v in model.full_state_dict().items()}
    teacher = BS.Teacher(features)
    rng = np.random.default_rng(5)
    B = 256 * world                            # global batch
    batches = BS.make_batches("c5_dlrm", spec, cards, rng, B, 4, "powerlaw", teacher)

    def part(b):
        lo, hi = rank * B // world, (rank + 1) * B // world
        return {k: torch.from_numpy(np.asarray(v)[lo:hi]) for k, v in b.items()}
    model.train()
    model._max_gradient_norm = 10.0
    losses = []
    for b in batches:
        loss = model.train_step(part(b)).detach().cpu().reshape(1).clone()
        dist.all_reduce(loss)
        losses.append(float(loss) / world)
    model.optimizer.check_errors()
    model.eval()
    with torch.no_grad():
        p = model.forward(part(batches[-1]))["y_pred"].reshape(-1).cpu()
    gp = [torch.empty_like(p) for _ in range(world)]
    dist.all_gather(gp, p)
    full = model.full_state_dict()
    if rank == 0:
        tr = O.OracleTrainer(cfg, state0, features, lr=1e-3, max_norm=10.0)
        ref_losses = [tr.train_step(BS.tb(b), BS.tb(b)["label"])[0] for b in batches]
        ref_p = tr.predict(BS.tb(batches[-1])).reshape(-1).numpy()
        wdiff = max(float((full[k].cpu() - tr.state[k].detach()).abs().max()) for k in full
                    if full[k].is_floating_point())
        np.savez(out_path, losses=np.asarray(losses), ref_losses=np.asarray(ref_losses),
                 pred=torch.cat(gp).numpy(), ref_pred=ref_p, wdiff=np.asarray([wdiff]))
    dist.barrier()
    _shutdown(model)


def run(rank, world, case, port, out_path, use_gpu):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    nccl = os.environ.get("FX_TEST_BACKEND") == "nccl"      # RCCL: one rank per device only
    if nccl:
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)

    def comm(t):            # tensors of the test's own bookkeeping collectives
        return t.cuda() if nccl else t
    from conftest import Golden, _din_fields
    if not use_gpu:
        import _cpu_emul
        _cpu_emul.install_plain()
    from fuxictr_amd import zoo
    from fuxictr_amd.features import FeatureMap
    g = Golden(case)
    m = g.meta
    fmap = FeatureMap(g.spec["dataset_id"], "/tmp")
    fmap.load_dict(g.spec, {"embedding_dim": m["embedding_dim"]})
    common = dict(gpu=0 if use_gpu else -1, embedding_dim=m["embedding_dim"],
                  learning_rate=m["lr"], optimizer=m["optimizer"], loss="binary_crossentropy",
                  task="binary_classification", metrics=["logloss", "AUC"], verbose=0,
                  model_root="/tmp/fx_dist_%d" % rank, shard="row",
                  embedding_regularizer=m.get("emb_reg", 0), net_regularizer=m.get("net_reg", 0),
                  hip_graph=os.environ.get("FX_HIP_GRAPH", "0") == "1")
    if m["model"] == "DeepFM":
        model = zoo.DeepFM(fmap, model_id=case, hidden_units=m["hidden"],
                           batch_norm=m.get("batch_norm", False), **common)
    elif m["model"] == "xDeepFM":
        model = zoo.xDeepFM(fmap, model_id=case, dnn_hidden_units=m["hidden"],
                            cin_hidden_units=m["cin"], **common)
    elif m["model"] == "DLRM":
        model = zoo.DLRM(fmap, model_id=case, top_mlp_units=m["hidden"],
                         bottom_mlp_units=m["bottom"], interaction_op=m.get("interaction_op", "dot"), **common)
    elif m["model"] == "DIN":
        model = zoo.DIN(fmap, model_id=case, dnn_hidden_units=m["hidden"],
                        dnn_activations="relu", attention_hidden_units=m["att_hidden"],
                        attention_hidden_activations="Dice", din_target_field=_din_fields(m, "din_target", "adgroup_id"),
                        din_sequence_field=_din_fields(m, "din_sequence", "click_sequence"),
                        din_use_softmax=m.get("din_softmax", False), **common)
    else:
        model = zoo.DCNv2(fmap, model_id=case, model_structure=m.get("structure", "parallel"),
                          num_cross_layers=m["n_cross"], parallel_dnn_hidden_units=m["hidden"], stacked_dnn_hidden_units=m.get("stacked", []),
                          **common)
    for grp_mod in model.modules():
        if hasattr(grp_mod, "table_groups"):
            for grp in grp_mod.table_groups():
                grp.a2a_factor = float(os.environ.get("FX_A2A_FACTOR", "1.5"))
    model.load_full_state_dict({k: torch.from_numpy(v) for k, v in g.state0.items()})
    model._max_gradient_norm = m["max_norm"]

    def part(b):
        n = len(b["label"])
        lo, hi = rank * n // world, (rank + 1) * n // world
        return {k: torch.from_numpy(np.asarray(v)[lo:hi]) for k, v in b.items()}

    if os.environ.get("FX_TEST_FIT") == "1":
        # BaseModel.fit across ranks: every rank feeds its shard of the batches and ITS shard of the
        # validation set; the evaluations must see the GLOBAL validation set on every rank so that
        # lr decay / early stop / best checkpoint agree (no rank may leave the loop alone)
        class Gen(list):
            pass
        train = Gen(part(b) for b in g.batches[:-1])
        valid = Gen([part(g.batches[-1])])
        model._monitor_mode = "max"
        model.fit(train, epochs=4, validation_data=valid, max_gradient_norm=m["max_norm"])
        logs = model.evaluate(valid)
        mine = torch.tensor([model._best_metric, float(model.optimizer.param_groups[0]["lr"]),
                             float(model._total_steps), float(model._stop_training),
                             float(model._epoch_index), logs["AUC"], logs["logloss"]],
                            dtype=torch.float64)
        allv = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allv, comm(mine))
        # the global-set metric, recomputed from the gathered shard predictions
        model.eval()
        with torch.no_grad():
            p = model.forward(part(g.batches[-1]))["y_pred"].reshape(-1).double().cpu()
        gp = [torch.empty_like(p) for _ in range(world)]
        dist.all_gather(gp, comm(p))
        if rank == 0:
            np.savez(out_path, fit=torch.stack([t.cpu() for t in allv]).numpy(),
                     pred=torch.cat([t.cpu() for t in gp]).numpy())
        dist.barrier()
        _shutdown(model)
        return
    model.eval()
    with torch.no_grad():
        p0 = model.forward(part(g.batches[-1]))["y_pred"].reshape(-1).cpu()
    model.train()
    losses = []
    for i in range(m["steps"]):
        loss = comm(model.train_step(part(g.batches[i])).detach().cpu().reshape(1).clone())
        dist.all_reduce(loss)
        losses.append(float(loss) / world)
    model.optimizer.check_errors()
    model.eval()
    with torch.no_grad():
        p1 = model.forward(part(g.batches[-1]))["y_pred"].reshape(-1).cpu()
    if os.environ.get("FX_TEST_CKPT") == "1":
        # per-shard checkpoint files: every rank writes / reads its own rows
        ck = os.path.join(os.path.dirname(out_path), "ck", "m.model")
        model.save_weights(ck)
        assert os.path.exists("%s.rank%d-of-%d" % (ck, rank, world))
        before = {k: v.clone() for k, v in model.state_dict().items()}
        with torch.no_grad():
            for p_ in model.parameters():
                p_.add_(1.0)
        model.load_weights(ck)
        for k, v in model.state_dict().items():
            assert torch.equal(v, before[k]), k
    full = {k: v.cpu().numpy() for k, v in model.full_state_dict().items()}
    p0, p1 = comm(p0), comm(p1)
    gp0 = [torch.empty_like(p0) for _ in range(world)]
    gp1 = [torch.empty_like(p1) for _ in range(world)]
    dist.all_gather(gp0, p0)
    dist.all_gather(gp1, p1)
    gp0, gp1 = [t.cpu() for t in gp0], [t.cpu() for t in gp1]
    if rank == 0:
        np.savez(out_path, sharded=np.asarray([model._dist is not None]),
                 losses=np.asarray(losses), pred0=torch.cat(gp0).numpy(),
                 pred1=torch.cat(gp1).numpy(), **{"state/" + k: v for k, v in full.items()})
    dist.barrier()
    _shutdown(model)


if __name__ == "__main__":
    rank, world, case, port, out_path, use_gpu = sys.argv[1:7]
    if case == "c5_dlrm":
        run_c5(int(rank), int(world), int(port), out_path)
    else:
        run(int(rank), int(world), case, int(port), out_path, use_gpu == "1")
