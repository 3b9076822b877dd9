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
