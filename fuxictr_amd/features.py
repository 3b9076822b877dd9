"""Feature schema registry — host-side mirror of the reference's `fuxictr.features.FeatureMap`
(fuxictr/features.py:26-196): same attributes, same `feature_map.json` format, same column
indexing, so the native layers consume exactly what the reference's preprocessing emits.
"""
import json
import logging
import os
from collections import OrderedDict


class FeatureMap(object):
    """name -> spec registry (type, source, vocab_size, padding_idx, share_embedding, ...).

    Mirrors fuxictr/features.py:26 (ctor :49-59, load :54-78, sum_emb_out_dim :134-154,
    set_column_index :156-180).
    """

    def __init__(self, dataset_id, data_dir):
        self.data_dir = data_dir
        self.dataset_id = dataset_id
        self.num_fields = 0
        self.total_features = 0
        self.input_length = 0
        self.features = OrderedDict()
        self.labels = []
        self.column_index = dict()
        self.group_id = None
        self.default_emb_dim = None

    # -- construction -------------------------------------------------------------------------
    def load(self, json_file, params):
        logging.info("Load feature_map from json: " + json_file)
        with open(json_file, "r", encoding="utf-8") as fd:
            spec = json.load(fd)
        self.load_dict(spec, params)

    def load_dict(self, spec, params):
        """Same as `load` for an already parsed feature_map dict (used by synthetic datasets)."""
        if spec["dataset_id"] != self.dataset_id:
            raise RuntimeError("dataset_id={} does not match feature_map!".format(self.dataset_id))
        self.labels = spec.get("labels", [])
        self.total_features = spec.get("total_features", 0)
        self.input_length = spec.get("input_length", 0)
        self.group_id = params.get("group_id", None)
        self.default_emb_dim = params.get("embedding_dim", None)
        self.features = OrderedDict((k, v) for item in spec["features"] for k, v in item.items())
        self.num_fields = self.get_num_fields()
        if params.get("use_features", None):
            self.features = OrderedDict((x, self.features[x]) for x in params["use_features"])
        if params.get("feature_specs", None):
            self.update_feature_specs(params["feature_specs"])
        self.set_column_index()

    def update_feature_specs(self, feature_specs):
        for col in feature_specs:
            names = col["name"]
            if not isinstance(names, list):
                names = [names]
            for name in names:
                for k, v in col.items():
                    if k != "name":
                        self.features[name][k] = v

    def save(self, json_file):
        logging.info("Save feature_map to json: " + json_file)
        os.makedirs(os.path.dirname(json_file), exist_ok=True)
        out = OrderedDict()
        out["dataset_id"] = self.dataset_id
        out["num_fields"] = self.num_fields
        out["total_features"] = self.total_features
        out["input_length"] = self.input_length
        out["labels"] = self.labels
        out["features"] = [{k: v} for k, v in self.features.items()]
        with open(json_file, "w") as fd:
            json.dump(out, fd, indent=4)

    # -- queries ------------------------------------------------------------------------------
    def get_num_fields(self, feature_source=[]):
        if not isinstance(feature_source, list):
            feature_source = [feature_source]
        n = 0
        for _, spec in self.features.items():
            if spec["type"] == "meta":
                continue
            if len(feature_source) == 0 or spec.get("source") in feature_source:
                n += 1
        return n

    def sum_emb_out_dim(self, feature_source=[]):
        if not isinstance(feature_source, list):
            feature_source = [feature_source]
        total = 0
        for _, spec in self.features.items():
            if spec["type"] == "meta":
                continue
            if len(feature_source) == 0 or spec.get("source") in feature_source:
                total += spec.get("emb_output_dim",
                                  spec.get("embedding_dim", self.default_emb_dim))
        return total

    def set_column_index(self):
        idx = 0
        for feature, spec in self.features.items():
            if spec["type"] == "sequence":
                self.column_index[feature] = [i + idx for i in range(spec["max_len"])]
                idx += spec["max_len"]
            elif spec["type"] == "embedding":
                dim = spec["pretrain_dim"]
                self.column_index[feature] = [i + idx for i in range(dim)]
                idx += dim
            else:
                self.column_index[feature] = idx
                idx += 1
        self.input_length = idx
        for label in self.labels:
            self.column_index[label] = idx
            idx += 1

    def get_column_index(self, feature):
        if feature not in self.column_index:
            self.set_column_index()
        return self.column_index[feature]
