"""The [p | m | v | last_step] row record (round 6, `_TableGroup.adopt_record`, struct fx_row_state's row strides)
on a real MI355X.  The record changes WHERE a row's four fields live, not one bit of what is computed:
  * a DeepFM / DIN model trained with the record equals the same model trained on four packed arrays
    (FX_ROW_RECORD=0 layout) bit for bit — losses, tables, moments, stamps — eagerly and under hipGraph replay;
  * the per-feature Parameters stay views of the one storage (state_dict keys and aliasing as the reference's);
  * save_weights writes compact tables (not the record's storage), checkpoints resume across layouts;
  * the stride-taking forward entry points (fx_emb_gather_fwd, fx_emb_seq_pool_fwd, fx_lr_fwd) read a table
    inside a wider block exactly as they read the packed copy."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from fuxictr_amd import ops, optim, synthetic, zoo  # noqa: E402
from fuxictr_amd.layers import _TableGroup  # noqa: E402
from test_gpu_fused import DEV, _dev, _ids, _schema  # noqa: E402


def tb(batch):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in batch.items()}


def _deepfm(tmp, scale=0.002, graph=False):
    cards = [max(3, int(c * scale)) for c in synthetic.CRITEO_CARDS]
    fmap, _ = synthetic.criteo_feature_map(cards=cards, embedding_dim=16)
    torch.manual_seed(3)
    model = zoo.DeepFM(fmap, model_id="rec", gpu=0, embedding_dim=16, hidden_units=[64, 32],
                       optimizer="adam", loss="binary_crossentropy", learning_rate=1e-3,
                       task="binary_classification", metrics=["logloss", "AUC"], verbose=0,
                       model_root=str(tmp), sparse_update="exact", hip_graph=graph)
    rng = np.random.default_rng(1)
    return model, [tb(synthetic.criteo_batch(rng, 512, cards=cards)) for _ in range(12)]


def _din(tmp, scale=0.002, graph=False):
    fmap, spec = synthetic.taobao_feature_map(max_len=12, embedding_dim=16, scale=scale)
    torch.manual_seed(4)
    model = zoo.DIN(fmap, model_id="recdin", gpu=0, embedding_dim=16, dnn_hidden_units=[64, 32],
                    attention_hidden_units=[32], din_target_field=["adgroup_id"],
                    din_sequence_field=["click_sequence"], optimizer="adam", loss="binary_crossentropy",
                    learning_rate=1e-3, task="binary_classification", metrics=["logloss", "AUC"], verbose=0,
                    model_root=str(tmp), sparse_update="exact", hip_graph=graph)
    rng = np.random.default_rng(2)
    return model, [tb(synthetic.taobao_batch(rng, 256, spec)) for _ in range(12)]


def _groups(model):
    out = []
    for mod in model.modules():
        if hasattr(mod, "table_groups"):
            out += [g for g in mod.table_groups() if g.table is not None]
    return out


def _run(make, tmp, record, graph):
    old = optim.ROW_RECORD
    optim.ROW_RECORD = record
    try:
        model, batches = make(tmp, graph=graph)
    finally:
        optim.ROW_RECORD = old
    model.train()
    losses = [float(model.train_step(b).item()) for b in batches]      # (graph: 3 eager, capture, replays)
    model.optimizer.check_errors()
    return model, losses


@pytest.mark.parametrize("graph", [False, True], ids=["eager", "hipgraph"])
@pytest.mark.parametrize("make", [_deepfm, _din], ids=["deepfm", "din"])
def test_record_layout_trains_bit_identically_to_four_packed_arrays(make, graph, tmp_path):
    ma, la = _run(make, tmp_path, True, graph)
    mb, lb = _run(make, tmp_path, False, graph)
    assert la == lb
    ga, gb = _groups(ma), _groups(mb)
    assert len(ga) == len(gb) and len(ga) >= 1
    for a, b in zip(ga, gb):
        assert a.record is not None and b.record is None
        W = _TableGroup.record_width(a.D)
        assert a.record.shape == (a.table.shape[0], W) and a.table.stride(0) == W
        assert a.table.data_ptr() == a.record.data_ptr()
        assert a.last_step.dtype == torch.int32 and a.last_step.stride(0) == W
        for name in ("table", "m", "v", "last_step"):
            assert torch.equal(getattr(a, name), getattr(b, name)), name
        assert int(a.last_step.max()) > 0
    # Parameters are views of the record; keys and values as in the packed model
    sa, sb = ma.state_dict(), mb.state_dict()
    assert list(sa.keys()) == list(sb.keys())
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    g0 = ga[0]
    f0, (base, V, _) = next(iter(g0.tables.items()))
    for mod in ma.modules():
        if hasattr(mod, "table_groups") and g0 in mod.table_groups():
            w = mod.embedding_layers[f0].weight
            assert w.data_ptr() == g0.table[base:base + V].data_ptr() and w.stride(0) == g0.table.stride(0)


def test_record_layout_checkpoints_are_compact_and_cross_load(tmp_path):
    ma, la = _run(_deepfm, tmp_path, True, False)
    mb, lb = _run(_deepfm, tmp_path, False, False)
    pa, pb = str(tmp_path / "a" / "m.model"), str(tmp_path / "b" / "m.model")
    ma.save_checkpoint(pa)
    mb.save_checkpoint(pb)
    assert abs(os.path.getsize(pa) - os.path.getsize(pb)) < 65536       # tables only, not the 4 x larger records
    # cross-load: the record model resumes from the packed model's checkpoint and vice versa
    _, batches = _deepfm(tmp_path)
    ma.load_checkpoint(pb)
    mb.load_checkpoint(pa)
    ma.train()
    mb.train()
    for b in batches[:4]:
        assert float(ma.train_step(b).item()) == float(mb.train_step(b).item())
    for a, b in zip(_groups(ma), _groups(mb)):
        for name in ("table", "m", "v", "last_step"):
            assert torch.equal(getattr(a, name), getattr(b, name)), name


def test_forward_entry_points_read_a_table_inside_a_wider_block():
    rng = np.random.default_rng(11)
    g = torch.Generator().manual_seed(11)
    D, B, W = 16, 700, 64
    vocabs = [50, 3, 1000, 7, 2011]
    bases, R = _schema(vocabs)
    C = len(vocabs)
    block = torch.randn(R, W, generator=g).to(DEV)
    wide, packed = block[:, :D], block[:, :D].contiguous()
    wide1, packed1 = block[:, 49:50], block[:, 49:50].contiguous()
    ids = _dev(_ids(rng, B, vocabs, "power"), torch.int32)
    off = _dev([c * D for c in range(C)], torch.int64)
    scal = ops.new_scalars(DEV)
    outs = []
    seq_ids = _dev(rng.integers(0, vocabs[2], (B, 6)), torch.int32)
    for t, t1 in ((wide, wide1), (packed, packed1)):
        out = torch.empty(B, C * D, device=DEV)
        ops.emb_gather_fwd(t, D, ids, _dev(bases, torch.int64), _dev(vocabs, torch.int32), off, None, None,
                           None, out, scal)
        lr = torch.empty(B, 1, device=DEV)
        ops.lr_fwd(t1, ids, _dev(bases, torch.int64), _dev(vocabs, torch.int32), None, None, None, lr, scal)
        # ONE mean-pooled sequence of length 6 over table 2's vocabulary
        pooled = torch.zeros(B, D, device=DEV)
        denom = torch.empty(B, 1, device=DEV)
        ops.emb_seq_pool_fwd(t, D, seq_ids, _dev([bases[2]] * 6, torch.int64), _dev([vocabs[2]] * 6, torch.int32),
                             _dev([0], torch.int32), _dev([6], torch.int32), _dev([ops.POOL_MEAN], torch.int32),
                             _dev([0], torch.int64), pooled, denom, scal)
        outs.append((out, lr, pooled, denom))
    torch.cuda.synchronize()
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    # and the gather equals direct indexing of the strided view
    rows = ids.long() + _dev(bases, torch.int64)[None, :]
    assert torch.equal(outs[0][0].view(B, C, D), wide[rows])
