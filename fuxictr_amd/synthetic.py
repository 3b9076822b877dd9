"""Synthetic datasets of the shapes BASELINE.json names: there is no network, so
Criteo itself is not available; these reproduce its schema and cardinalities."""
import numpy as np

from .features import FeatureMap

# Criteo-Kaggle categorical cardinalities (sum = 33 762 577), SURVEY.md §8(d)
CRITEO_CARDS = [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194,
                27, 14992, 5461306, 10, 5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572]


def criteo_feature_map(cards=None, n_dense=13, embedding_dim=16, dataset_id="synthetic_criteo"):
    """13 numeric I1..I13 + 26 categorical C1..C26, padding_idx 0, vocab_size = card + 1."""
    cards = list(CRITEO_CARDS if cards is None else cards)
    feats = []
    for j in range(n_dense):
        feats.append({"I%d" % (j + 1): {"source": "", "type": "numeric"}})
    for c, card in enumerate(cards):
        feats.append({"C%d" % (c + 1): {"source": "", "type": "categorical", "padding_idx": 0,
                                       "vocab_size": int(card) + 1}})
    spec = {"dataset_id": dataset_id, "num_fields": len(feats),
            "total_features": int(sum(cards)) + len(cards), "input_length": len(feats),
            "labels": ["label"], "features": feats}
    fmap = FeatureMap(dataset_id, data_dir="")
    fmap.load_dict(spec, {"embedding_dim": embedding_dim})
    return fmap, spec


def criteo_batch(rng, B, cards=None, n_dense=13, dist="powerlaw", label_rate=0.25):
    """One batch as numpy columns: ids int64 in [1, card] (`uniform`, or `powerlaw`
    floor(card * u^3) + 1), dense fp32 U[0,1), label Bernoulli(label_rate) as float32."""
    cards = np.asarray(CRITEO_CARDS if cards is None else cards, dtype=np.int64)
    out = {}
    for j in range(n_dense):
        out["I%d" % (j + 1)] = rng.random(B, dtype=np.float32)
    for c, card in enumerate(cards):
        u = rng.random(B)
        if dist == "uniform":
            ids = np.floor(u * card).astype(np.int64) + 1
        elif dist == "powerlaw":
            ids = np.floor(card * u ** 3).astype(np.int64) + 1
        else:
            raise ValueError("dist=%s" % dist)
        out["C%d" % (c + 1)] = np.minimum(ids, card)
    out["label"] = (rng.random(B) < label_rate).astype(np.float32)
    return out


# Taobao-ad-shaped sequence data for DIN (BASELINE config c4)
TAOBAO_CARDS = [("userid", 1141729), ("adgroup_id", 846811), ("pid", 2), ("cate_id", 6769),
                ("campaign_id", 423436), ("customer", 255875), ("brand", 99815),
                ("cms_segid", 97), ("cms_group_id", 13), ("final_gender_code", 2),
                ("age_level", 7), ("pvalue_level", 4), ("shopping_level", 3), ("occupation", 2)]


def taobao_feature_map(max_len=50, embedding_dim=16, dataset_id="synthetic_taobao", scale=1.0):
    feats = []
    for name, card in TAOBAO_CARDS:
        card = max(2, int(card * scale))
        feats.append({name: {"source": "", "type": "categorical", "padding_idx": 0,
                             "vocab_size": card + 1}})
    item_card = max(2, int(dict(TAOBAO_CARDS)["adgroup_id"] * scale))
    feats.append({"click_sequence": {"source": "", "type": "sequence", "feature_encoder": None,
                                     "share_embedding": "adgroup_id", "padding_idx": 0,
                                     "vocab_size": item_card + 1, "max_len": max_len}})
    spec = {"dataset_id": dataset_id, "num_fields": len(feats), "total_features": 0,
            "input_length": 0, "labels": ["label"], "features": feats}
    fmap = FeatureMap(dataset_id, data_dir="")
    fmap.load_dict(spec, {"embedding_dim": embedding_dim})
    return fmap, spec


def taobao_batch(rng, B, spec, dist="powerlaw", label_rate=0.05):
    out = {}
    for item in spec["features"]:
        (name, fs), = item.items()
        card = fs["vocab_size"] - 1
        if fs["type"] == "sequence":
            L = fs["max_len"]
            u = rng.random((B, L))
            ids = (np.floor(card * u ** 3) if dist == "powerlaw" else np.floor(card * u))
            ids = np.minimum(ids.astype(np.int64) + 1, card)
            lens = rng.integers(1, L + 1, size=B)
            ids[np.arange(L)[None, :] >= lens[:, None]] = 0      # post-padded with 0
            out[name] = ids
        else:
            u = rng.random(B)
            ids = (np.floor(card * u ** 3) if dist == "powerlaw" else np.floor(card * u))
            out[name] = np.minimum(ids.astype(np.int64) + 1, card)
    out["label"] = (rng.random(B) < label_rate).astype(np.float32)
    return out
