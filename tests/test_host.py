"""CPU checks of the host-side mirror: FeatureMap, table packing / state_dict layout of the native
FeatureEmbeddingDict (storage only — no kernel is called), synthetic data."""
import json

import numpy as np
import pytest
import torch

from fuxictr_amd import layers, synthetic
from fuxictr_amd.features import FeatureMap


def _fmap(tmp_path, spec, params=None):
    p = tmp_path / "feature_map.json"
    p.write_text(json.dumps(spec))
    fm = FeatureMap(spec["dataset_id"], str(tmp_path))
    fm.load(str(p), params or {"embedding_dim": 8})
    return fm


SEQ_SPEC = {
    "dataset_id": "t", "num_fields": 4, "total_features": 0, "input_length": 0, "labels": ["clk"],
    "features": [
        {"price": {"source": "item", "type": "numeric"}},
        {"adgroup_id": {"source": "item", "type": "categorical", "padding_idx": 0, "vocab_size": 20}},
        {"userid": {"source": "user", "type": "categorical", "padding_idx": 0, "vocab_size": 7}},
        {"click_sequence": {"source": "user", "type": "sequence", "padding_idx": 0,
                            "vocab_size": 20, "max_len": 5, "share_embedding": "adgroup_id",
                            "feature_encoder": None}},
        {"gid": {"type": "meta"}},
    ]}


def test_feature_map_mirror(tmp_path):
    fm = _fmap(tmp_path, SEQ_SPEC)
    assert fm.num_fields == 4 and fm.labels == ["clk"]
    assert fm.sum_emb_out_dim() == 32
    assert fm.column_index["click_sequence"] == [3, 4, 5, 6, 7]
    assert fm.column_index["clk"] == 9 and fm.input_length == 9
    assert fm.get_num_fields("user") == 2
    with pytest.raises(RuntimeError):
        FeatureMap("other", str(tmp_path)).load(str(tmp_path / "feature_map.json"), {})
    fm.save(str(tmp_path / "out" / "fm.json"))
    assert json.loads((tmp_path / "out" / "fm.json").read_text())["dataset_id"] == "t"


def test_packed_table_layout_and_state_dict_keys(tmp_path):
    fm = _fmap(tmp_path, SEQ_SPEC)
    layers.set_default_device("cpu")
    try:
        emb = layers.FeatureEmbedding(fm, 8)
    finally:
        layers.set_default_device(None)
    d = emb.embedding_layer
    sd = emb.state_dict()
    pre = "embedding_layer.embedding_layers."
    assert sorted(sd.keys()) == sorted(pre + f + ".weight" for f in
                                       ["price", "adgroup_id", "userid", "click_sequence"])
    assert sd[pre + "price.weight"].shape == (8, 1)
    assert sd[pre + "adgroup_id.weight"].shape == (20, 8)
    # share_embedding: one Parameter under two keys, like the reference
    assert d.embedding_layers["click_sequence"] is d.embedding_layers["adgroup_id"]
    grp = d.table_groups()[0]
    assert grp.total_rows == 27 and grp.table.shape == (27, 8)
    # Parameters are views of the packed storage
    w = d.embedding_layers["userid"].weight
    assert w.data_ptr() == grp.table[20:].data_ptr()
    with torch.no_grad():
        w[3].fill_(5.0)
    assert float(grp.table[23, 0]) == 5.0
    # padding rows are zero, other rows ~ N(0, 1e-4)
    assert float(grp.table[0].abs().sum()) == 0 and float(grp.table[20].abs().sum()) == 0
    assert 0 < float(grp.table[1:20].std()) < 1e-3
    # plan: slots follow feature_map order, sequence takes max_len slots
    plan = grp.plan_for(["price", "adgroup_id", "userid", "click_sequence"])
    assert plan.n_slots == 8 and plan.C == 7 and plan.Fd == 1
    assert plan.col_row_base.tolist() == [0, 20, 0, 0, 0, 0, 0]
    assert plan.col_out_off.tolist() == [8, 16, 24, 32, 40, 48, 56]
    assert plan.num_out_off.tolist() == [0]
    # load_state_dict writes through the views
    new = {k: torch.full_like(v, 2.0) for k, v in sd.items()}
    emb.load_state_dict(new)
    assert float(grp.table.min()) == 2.0 and float(grp.num_w.min()) == 2.0


def test_lr_layer_tables_are_d1_and_unshared(tmp_path):
    fm = _fmap(tmp_path, SEQ_SPEC)
    layers.set_default_device("cpu")
    try:
        lr = layers.LogisticRegression(fm)
    finally:
        layers.set_default_device(None)
    d = lr.embedding_layer.embedding_layer
    grp = d.table_groups()[0]
    assert grp.D == 1 and grp.total_rows == 47          # click_sequence gets its own table
    assert d.embedding_layers["click_sequence"] is not d.embedding_layers["adgroup_id"]
    assert "click_sequence" in d.feature_encoders      # MaskedSumPooling, as the reference
    assert sorted(lr.state_dict().keys())[0] == "bias"


def test_native_ops_refuse_cpu_tensors():
    from fuxictr_amd import _lib, ops
    with pytest.raises(_lib.FxError):
        ops.pack_columns([torch.zeros(4)], torch.zeros(4, 1, dtype=torch.int32))


def test_pretrained_and_regularizer_are_explicit_gaps(tmp_path):
    spec = json.loads(json.dumps(SEQ_SPEC))
    spec["features"][1]["adgroup_id"]["pretrained_emb"] = "x.h5"
    fm = _fmap(tmp_path, spec)
    layers.set_default_device("cpu")
    try:
        with pytest.raises(NotImplementedError):
            layers.FeatureEmbedding(fm, 8)
    finally:
        layers.set_default_device(None)


def test_synthetic_criteo_shape():
    fmap, spec = synthetic.criteo_feature_map()
    assert fmap.num_fields == 39 and fmap.sum_emb_out_dim() == 624
    assert sum(synthetic.CRITEO_CARDS) == 33762577
    rng = np.random.default_rng(0)
    b = synthetic.criteo_batch(rng, 64, dist="powerlaw")
    assert b["C3"].min() >= 1 and b["C3"].max() <= 10131227 and b["I1"].dtype == np.float32
    u = synthetic.criteo_batch(rng, 64, dist="uniform")
    assert u["C9"].max() <= 3
