"""Feature schema registry — host-side mirror of the reference's `fuxictr.features.FeatureMap`
(fuxictr/features.py:26-196): same attributes, same `feature_map.json` format, same column
indexing, so the native layers consume exactly what the reference's preprocessing emits.
"""
import json
import logging
import os
from collections import OrderedDict

_JSON_FIELDS = ("dataset_id", "num_fields", "total_features", "input_length", "labels")


def _as_list(x):
    return x if isinstance(x, list) else [x]


class FeatureMap(object):
    """name -> spec registry (type, source, vocab_size, padding_idx, share_embedding, ...).

    Public surface as fuxictr/features.py:26 (ctor :49-59, load :54-78, sum_emb_out_dim :134-154,
    set_column_index :156-180); `load_dict` is an addition for schemas built in memory.
    """

    def __init__(self, dataset_id, data_dir):
        self.dataset_id, self.data_dir = dataset_id, data_dir
        self.features, self.labels, self.column_index = OrderedDict(), [], {}
        self.num_fields = self.total_features = self.input_length = 0
        self.group_id = self.default_emb_dim = None

    # -- construction -------------------------------------------------------------------------
    def load(self, json_file, params):
        logging.info("reading feature map %s", json_file)
        with open(json_file, encoding="utf-8") as fd:
            self.load_dict(json.load(fd), params)

    def load_dict(self, spec, params):
        """`load` for an already parsed feature_map dict (synthetic datasets, tests)."""
        if spec["dataset_id"] != self.dataset_id:
            raise RuntimeError("dataset_id={} does not match feature_map!".format(self.dataset_id))
        self.labels = spec.get("labels", [])
        self.total_features = spec.get("total_features", 0)
        self.input_length = spec.get("input_length", 0)
        self.group_id = params.get("group_id")
        self.default_emb_dim = params.get("embedding_dim")
        # the json stores one single-key dict per feature, in schema order
        self.features = OrderedDict(kv for entry in spec["features"] for kv in entry.items())
        self.num_fields = self.get_num_fields()      # counted BEFORE any use_features selection
        selected = params.get("use_features")
        if selected:
            self.features = OrderedDict((name, self.features[name]) for name in selected)
        overrides = params.get("feature_specs")
        if overrides:
            self.update_feature_specs(overrides)
        self.set_column_index()

    def update_feature_specs(self, feature_specs):
        """[{name: str | [str], key: value, ...}] -> patch the named features' specs."""
        for patch in feature_specs:
            changes = {k: v for k, v in patch.items() if k != "name"}
            for name in _as_list(patch["name"]):
                self.features[name].update(changes)

    def save(self, json_file):
        logging.info("writing feature map %s", json_file)
        os.makedirs(os.path.dirname(json_file), exist_ok=True)
        doc = OrderedDict((field, getattr(self, field)) for field in _JSON_FIELDS)
        doc["features"] = [{name: spec} for name, spec in self.features.items()]
        with open(json_file, "w") as fd:
            json.dump(doc, fd, indent=4)

    # -- queries ------------------------------------------------------------------------------
    def _selected(self, feature_source):
        """Specs of the non-meta features whose `source` is in feature_source ([] = all)."""
        wanted = _as_list(feature_source)
        return [spec for spec in self.features.values()
                if spec["type"] != "meta" and (not wanted or spec.get("source") in wanted)]

    def get_num_fields(self, feature_source=[]):
        return len(self._selected(feature_source))

    def sum_emb_out_dim(self, feature_source=[]):
        return sum(spec.get("emb_output_dim", spec.get("embedding_dim", self.default_emb_dim))
                   for spec in self._selected(feature_source))

    def set_column_index(self):
        """Column positions in the stacked data matrix: a sequence takes max_len columns, a
        pretrained-embedding input pretrain_dim, everything else one; labels follow the features."""
        cursor = 0
        for name, spec in self.features.items():
            width = {"sequence": spec.get("max_len"), "embedding": spec.get("pretrain_dim")}.get(
                spec["type"])
            if width is None:
                self.column_index[name] = cursor
                cursor += 1
            else:
                self.column_index[name] = list(range(cursor, cursor + width))
                cursor += width
        self.input_length = cursor
        for offset, label in enumerate(self.labels):
            self.column_index[label] = cursor + offset

    def get_column_index(self, feature):
        if feature not in self.column_index:
            self.set_column_index()
        return self.column_index[feature]
