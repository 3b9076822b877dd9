"""The callers of the hot path that BASELINE.json's configs name — DeepFM, DCNv2, DIN, DLRM, xDeepFM —
on the native drop-in layers.  Constructor keywords, attribute (= state_dict) names and the forward
composition are the reference's (model_zoo/DeepFM/DeepFM_torch/src/DeepFM.py:41-88,
model_zoo/DCNv2/src/DCNv2.py:44-132, model_zoo/DIN/src/DIN.py:50-150, model_zoo/DLRM/src/DLRM.py:44-124,
model_zoo/xDeepFM/src/xDeepFM.py:41-97), so its checkpoints and YAML configs apply unchanged.  These
classes exist because /root/reference does not travel to the GPU box; with the reference installed,
its own model_zoo classes run unmodified on the same layers through `fuxictr_amd.patch.install()`
(tests/test_dropin_reference_zoo.py).
"""
import os as _os

import torch
from torch import nn

from .layers import (CompressedInteractionNet, CrossNetV2, DIN_Attention, Dice,
                     FactorizationMachine, FeatureEmbedding, FeatureEmbeddingDict, FxLinear,
                     InnerProductInteraction, LogisticRegression, MLP_Block, _DlrmMixFn, _MLP_PAD,
                     _RecordGradSlot, din_record_layout)
from .rank_model import BaseModel


class _ZooModel(BaseModel):
    """Shared plumbing of the five models: base-class construction and the closing
    compile / reset_parameters / model_to_device sequence every model_zoo ctor ends with."""

    def _base(self, feature_map, model_id, gpu, embedding_regularizer, net_regularizer, kwargs):
        BaseModel.__init__(self, feature_map, model_id=model_id, gpu=gpu,
                           embedding_regularizer=embedding_regularizer,
                           net_regularizer=net_regularizer, **kwargs)

    def _ready(self, kwargs, learning_rate):
        self.compile(kwargs["optimizer"], kwargs["loss"], learning_rate)
        self.reset_parameters()
        self.model_to_device()

    def _tower(self, input_dim, units, activations, dropout, batch_norm, output_dim=1,
               output_activation=None):
        return MLP_Block(input_dim=input_dim, output_dim=output_dim, hidden_units=units,
                         hidden_activations=activations, output_activation=output_activation,
                         dropout_rates=dropout, batch_norm=batch_norm)


class DeepFM(_ZooModel):
    def __init__(self, feature_map, model_id="DeepFM", gpu=-1, learning_rate=1e-3,
                 embedding_dim=10, hidden_units=[64, 64, 64], hidden_activations="ReLU",
                 net_dropout=0, batch_norm=False, embedding_regularizer=None,
                 net_regularizer=None, **kwargs):
        self._base(feature_map, model_id, gpu, embedding_regularizer, net_regularizer, kwargs)
        self.embedding_layer = FeatureEmbedding(feature_map, embedding_dim)
        self.fm = FactorizationMachine(feature_map)
        self.mlp = self._tower(feature_map.sum_emb_out_dim(), hidden_units, hidden_activations,
                               net_dropout, batch_norm)
        self._ready(kwargs, learning_rate)

    def forward(self, inputs):
        X = self.get_inputs(inputs)
        emb = self.embedding_layer(X)                       # [B, F, D]
        logit = self.fm(X, emb)                             # first order + FM second order
        logit = self.mlp(emb.flatten(start_dim=1), out_add=logit)   # += in the head GEMM's epilogue
        return {"y_pred": self.output_activation(logit)}


class DCNv2(_ZooModel):
    _STRUCTURES = {"crossnet_only": (False, False), "stacked": (True, False),
                   "parallel": (False, True), "stacked_parallel": (True, True)}

    def __init__(self, feature_map, model_id="DCNv2", gpu=-1, model_structure="parallel",
                 use_low_rank_mixture=False, low_rank=32, num_experts=4, learning_rate=1e-3,
                 embedding_dim=10, stacked_dnn_hidden_units=[], parallel_dnn_hidden_units=[],
                 dnn_activations="ReLU", num_cross_layers=3, net_dropout=0, batch_norm=False,
                 embedding_regularizer=None, net_regularizer=None, **kwargs):
        self._base(feature_map, model_id, gpu, embedding_regularizer, net_regularizer, kwargs)
        assert model_structure in self._STRUCTURES, \
            "model_structure={} not supported!".format(model_structure)
        if use_low_rank_mixture:
            raise NotImplementedError("CrossNetMix is outside the hot-path scope (SURVEY §2 #5)")
        self.model_structure = model_structure
        stacked, parallel = self._STRUCTURES[model_structure]
        width = feature_map.sum_emb_out_dim()
        self.embedding_layer = FeatureEmbedding(feature_map, embedding_dim)
        self.crossnet = CrossNetV2(width, num_cross_layers)
        head_in = 0 if stacked else width                   # what feeds `fc` from the cross branch
        if stacked:                                         # cross output -> DNN
            self.stacked_dnn = self._tower(width, stacked_dnn_hidden_units, dnn_activations,
                                           net_dropout, batch_norm, output_dim=None)
            head_in = stacked_dnn_hidden_units[-1]
        if parallel:                                        # embeddings -> DNN, next to the cross
            self.parallel_dnn = self._tower(width, parallel_dnn_hidden_units, dnn_activations,
                                            net_dropout, batch_norm, output_dim=None)
            head_in += parallel_dnn_hidden_units[-1]
        self.fc = FxLinear(head_in, 1, device=self.device)
        self._ready(kwargs, learning_rate)

    def forward(self, inputs):
        X = self.get_inputs(inputs)
        flat = self.embedding_layer(X, flatten_emb=True)    # [B, F*D]
        stacked, parallel = self._STRUCTURES[self.model_structure]
        fused = self._fused_parallel(flat) if (parallel and not stacked) else None
        if fused is not None:
            branch = fused                                   # [cross | deep], one buffer, no cat
        else:
            branch = self.crossnet(flat)
            if stacked:
                branch = self.stacked_dnn(branch)
            if parallel:
                branch = torch.cat([branch, self.parallel_dnn(flat)], dim=-1)
        return {"y_pred": self.output_activation(self.fc(branch))}

    def _fused_parallel(self, flat):
        """`parallel`: cross layer i and deep layer i as one grid (layers._CrossDeepFn) when the deep
        tower is a plain Linear / ReLU stack over 16-byte aligned rows; else None."""
        from .layers import _CrossDeepFn
        fz = getattr(self.parallel_dnn, "_fused", None)
        if fz is None or fz[1] or flat.dim() != 2 or flat.shape[1] % 4 or self.crossnet.num_layers < 1:
            return None
        stack = fz[0]
        if any(lin.weight.shape[0] % 4 for lin, _ in stack):
            return None
        wb = []
        for lin in self.crossnet.cross_layers:
            wb += [lin.weight, lin.bias]
        for lin, _ in stack:
            wb += [lin.weight, lin.bias]
        return _CrossDeepFn.apply(flat, self.crossnet.num_layers, tuple(r for _, r in stack), *wb)


def _fields(spec):
    """'a' | ('a','b') | ['a','b'] -> tuple of field names."""
    return tuple(spec) if isinstance(spec, (list, tuple)) else (spec,)


class DIN(_ZooModel):
    def __init__(self, feature_map, model_id="DIN", gpu=-1, dnn_hidden_units=[512, 128, 64],
                 dnn_activations="ReLU", attention_hidden_units=[64],
                 attention_hidden_activations="Dice", attention_output_activation=None,
                 attention_dropout=0, learning_rate=1e-3, embedding_dim=10, net_dropout=0,
                 batch_norm=False, din_target_field=[("item_id", "cate_id")],
                 din_sequence_field=[("click_history", "cate_history")], din_use_softmax=False,
                 embedding_regularizer=None, net_regularizer=None, **kwargs):
        self._base(feature_map, model_id, gpu, embedding_regularizer, net_regularizer, kwargs)
        targets = din_target_field if isinstance(din_target_field, list) else [din_target_field]
        sequences = din_sequence_field if isinstance(din_sequence_field, list) \
            else [din_sequence_field]
        assert len(targets) == len(sequences), "len(din_target_field) != len(din_sequence_field)"
        # the public attributes keep the reference's shape: a str, or a tuple for grouped fields
        self.din_target_field = [tuple(f) if isinstance(f, list) else f for f in targets]
        self.din_sequence_field = [tuple(f) if isinstance(f, list) else f for f in sequences]
        self.embedding_dim = embedding_dim
        self.embedding_layer = FeatureEmbeddingDict(feature_map, embedding_dim)
        self.attention_layers = nn.ModuleList(
            DIN_Attention(embedding_dim * len(_fields(t)), attention_units=attention_hidden_units,
                          hidden_activations=attention_hidden_activations,
                          output_activation=attention_output_activation,
                          dropout_rate=attention_dropout, use_softmax=din_use_softmax)
            for t in self.din_target_field)
        if isinstance(dnn_activations, str) and dnn_activations.lower() == "dice":
            dnn_activations = [Dice(units) for units in dnn_hidden_units]
        self.dnn = self._tower(feature_map.sum_emb_out_dim(), dnn_hidden_units, dnn_activations,
                               net_dropout, batch_norm, output_activation=self.output_activation)
        # the reference's configuration — one target field, one raw sequence that is the LAST feature —
        # runs the attention inside the gather record (layers._DinRecordFn): no cat / slice / add launches
        self._in_record = None
        names = list(feature_map.features)
        if (_os.environ.get("FX_DIN_INPLACE", "1") != "0" and len(self.din_target_field) == 1
                and isinstance(self.din_target_field[0], str)
                and isinstance(self.din_sequence_field[0], str)
                and names and names[-1] == self.din_sequence_field[0]
                and feature_map.features[names[-1]]["type"] == "sequence"
                and not feature_map.features[names[-1]].get("feature_encoder")):
            self._in_record = (self.din_target_field[0], self.din_sequence_field[0])
            self.embedding_layer.reserve_pooled_slot(self.din_sequence_field[0])
        self._ready(kwargs, learning_rate)

    def _record_layout(self, emb):
        """(rec, tslot, hole) when the embedding dict is ONE gather record laid out
        [single-slot fields.. | reserved | the sequence's positions]; None otherwise."""
        if self._in_record is None:
            return None
        return din_record_layout(emb, *self._in_record)

    def get_embedding(self, field, feature_emb_dict):
        names = _fields(field)
        if len(names) == 1:
            return feature_emb_dict[names[0]]
        return torch.cat([feature_emb_dict[f] for f in names], dim=-1)

    def forward(self, inputs):
        X = self.get_inputs(inputs)
        emb = self.embedding_layer(X)                       # name -> [B,D] | [B,L,D]
        layout = self._record_layout(emb)
        if layout is not None:
            rec, tslot, hole = layout
            B, n_slots, D = rec.shape
            slot = _RecordGradSlot(B, n_slots, D, hole + 1, rec.device) if rec.requires_grad else None
            flat = self.attention_layers[0].forward_in_record(
                rec, self.embedding_layer.packed_ids(X, self._in_record[1]), tslot, hole, slot)
            if flat is not None:
                return {"y_pred": self.dnn(flat, dx_into=slot)}
        for attend, target, sequence in zip(self.attention_layers, self.din_target_field,
                                            self.din_sequence_field):
            seq_names = _fields(sequence)
            # padding_idx 0 marks the empty positions (DIN.py:125: `X[field].long() != 0`); the packed
            # int32 id columns themselves serve as the mask (kept when != 0): no cast / compare launches
            valid = self.embedding_layer.packed_ids(X, seq_names[0])
            if valid is None:
                valid = X[seq_names[0]].long() != 0
            pooled = attend(self.get_embedding(target, emb), self.get_embedding(sequence, emb),
                            valid)
            # the attended vector replaces each sequence field's [B,L,D] entry by its [B,D] share
            for name, part in zip(seq_names, pooled.split(self.embedding_dim, dim=-1)):
                emb[name] = part
        return {"y_pred": self.dnn(self.embedding_layer.dict2tensor(emb, flatten_emb=True))}


class DLRM(_ZooModel):
    def __init__(self, feature_map, model_id="DLRM", gpu=-1, learning_rate=1e-3, embedding_dim=10,
                 top_mlp_units=[64, 64, 64], bottom_mlp_units=[64, 64, 64],
                 top_mlp_activations="ReLU", bottom_mlp_activations="ReLU", top_mlp_dropout=0,
                 bottom_mlp_dropout=0, interaction_op="dot", batch_norm=False,
                 embedding_regularizer=None, net_regularizer=None, **kwargs):
        self._base(feature_map, model_id, gpu, embedding_regularizer, net_regularizer, kwargs)
        if interaction_op not in ("dot", "cat"):
            raise ValueError("interaction_op={} not supported.".format(interaction_op))
        self.interaction_op = interaction_op
        self.dense_feats = [name for name, spec in feature_map.features.items()
                            if spec["type"] == "numeric"]
        has_dense = len(self.dense_feats) > 0
        self.embedding_layer = FeatureEmbedding(feature_map, embedding_dim,
                                                not_required_feature_columns=self.dense_feats)
        # the numeric features enter as ONE extra "field": the bottom MLP's embedding_dim output
        n_fields = feature_map.num_fields - len(self.dense_feats) + int(has_dense)
        if has_dense:
            self.bottom_mlp = self._tower(len(self.dense_feats), bottom_mlp_units,
                                          bottom_mlp_activations, bottom_mlp_dropout, batch_norm,
                                          output_dim=embedding_dim,
                                          output_activation=bottom_mlp_activations)
        if interaction_op == "dot":
            self.interact = InnerProductInteraction(num_fields=n_fields, output="inner_product")
            top_in = n_fields * (n_fields - 1) // 2 + embedding_dim * int(has_dense)
        else:
            self.interact = nn.Flatten(start_dim=1)
            top_in = n_fields * embedding_dim
        self.top_mlp = self._tower(top_in, top_mlp_units, top_mlp_activations, top_mlp_dropout,
                                   batch_norm, output_activation=self.output_activation)
        # native fast path of the reference's configuration (dot interaction + numeric features): the
        # bottom tower's vector becomes the last field of the gather record (layers._DlrmMixFn)
        self._in_record = False
        if (_os.environ.get("FX_DLRM_INPLACE", "1") != "0" and has_dense and interaction_op == "dot"
                and n_fields <= 32 and embedding_dim <= 32 and embedding_dim % 2 == 0):
            self._in_record = True
            self.embedding_layer.embedding_layer.reserve_tail_slots(1)
        self._ready(kwargs, learning_rate)

    def _forward_in_record(self, X, fields):
        """-> the top tower's input, or None when this forward cannot take the in-record path."""
        records = getattr(fields, "_fx_records", None)
        if not self._in_record or not records or len(records) != 1:
            return None
        rec, plan = records[0]
        B, n_slots, D = rec.shape
        if plan.tail0 != n_slots - 1 or fields.shape[1] != n_slots - 1 \
                or getattr(self.bottom_mlp, "_fused", None) is None or self.bottom_mlp._fused[1] \
                or self.bottom_mlp._fused[0][-1][0].out_features != D:
            return None
        groups = list(self.embedding_layer.embedding_layer._groups.values())
        dense_in = groups[0].pack_dense(X, self.dense_feats)     # one cast launch (none in a captured step)
        dest = rec.detach()[:, n_slots - 1, :]
        dense_vec = self.bottom_mlp(dense_in, out_into=dest)       # written into the record's last slot
        assert dense_vec.data_ptr() == dest.data_ptr()
        P = n_slots * (n_slots - 1) // 2
        # the top tower pads an unaligned input width itself (one cat): hand it the padded row instead
        fused_top = getattr(self.top_mlp, "_fused", None) is not None
        pad = (-(P + D)) % 4 if (fused_top and P + D >= 64 and _MLP_PAD) else 0
        return _DlrmMixFn.apply(rec, dense_vec, pad)

    def forward(self, inputs):
        X = self.get_inputs(inputs)
        fields = self.embedding_layer(X)                    # [B, Fs, D]
        mixed = self._forward_in_record(X, fields)
        if mixed is not None:
            return {"y_pred": self.top_mlp(mixed)}
        dense_vec = None
        if self.dense_feats:
            dense_in = torch.cat([X[name].float().view(-1, 1) for name in self.dense_feats], dim=-1)
            dense_vec = self.bottom_mlp(dense_in)           # [B, D]
            fields = torch.cat([fields, dense_vec.unsqueeze(1)], dim=1)
        mixed = self.interact(fields)
        if dense_vec is not None and self.interaction_op == "dot":
            mixed = torch.cat([mixed, dense_vec], dim=-1)
        return {"y_pred": self.top_mlp(mixed)}


class xDeepFM(_ZooModel):
    def __init__(self, feature_map, model_id="xDeepFM", gpu=-1, learning_rate=1e-3,
                 embedding_dim=10, dnn_hidden_units=[64, 64, 64], dnn_activations="ReLU",
                 cin_hidden_units=[16, 16, 16], net_dropout=0, batch_norm=False,
                 embedding_regularizer=None, net_regularizer=None, **kwargs):
        self._base(feature_map, model_id, gpu, embedding_regularizer, net_regularizer, kwargs)
        self.embedding_layer = FeatureEmbedding(feature_map, embedding_dim)
        self.dnn = None
        if dnn_hidden_units:
            self.dnn = self._tower(feature_map.sum_emb_out_dim(), dnn_hidden_units,
                                   dnn_activations, net_dropout, batch_norm)
        self.lr_layer = LogisticRegression(feature_map, use_bias=False)
        self.cin = CompressedInteractionNet(feature_map.num_fields, cin_hidden_units, output_dim=1)
        self._ready(kwargs, learning_rate)

    def forward(self, inputs):
        X = self.get_inputs(inputs)
        emb = self.embedding_layer(X)
        # linear part + explicit interactions (+ implicit ones): the sums ride in the GEMM epilogues
        logit = self.cin(emb, out_add=self.lr_layer(X))
        if self.dnn is not None:
            logit = self.dnn(emb.flatten(start_dim=1), out_add=logit)
        return {"y_pred": self.output_activation(logit)}
