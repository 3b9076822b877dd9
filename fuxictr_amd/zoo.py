"""The callers of the hot path that BASELINE.json's configs name — DeepFM and DCNv2 — written
against the native drop-in layers exactly as the reference's model_zoo writes them against
`fuxictr.pytorch.layers` (model_zoo/DeepFM/DeepFM_torch/src/DeepFM.py:41-88,
model_zoo/DCNv2/src/DCNv2.py:44-132).  They exist here because /root/reference does not travel to
the GPU box; with the reference installed, its own model_zoo classes run unmodified on these
layers through `fuxictr_amd.patch.install()`.
"""
import torch
from torch import nn

from .layers import (CompressedInteractionNet, CrossNetV2, DIN_Attention, Dice,
                     FactorizationMachine, FeatureEmbedding, FeatureEmbeddingDict, FxLinear,
                     InnerProductInteraction, LogisticRegression, MLP_Block)
from .rank_model import BaseModel


class DeepFM(BaseModel):
    def __init__(self, feature_map, model_id="DeepFM", gpu=-1, learning_rate=1e-3,
                 embedding_dim=10, hidden_units=[64, 64, 64], hidden_activations="ReLU",
                 net_dropout=0, batch_norm=False, embedding_regularizer=None,
                 net_regularizer=None, **kwargs):
        super(DeepFM, self).__init__(feature_map, model_id=model_id, gpu=gpu,
                                     embedding_regularizer=embedding_regularizer,
                                     net_regularizer=net_regularizer, **kwargs)
        self.embedding_layer = FeatureEmbedding(feature_map, embedding_dim)
        self.fm = FactorizationMachine(feature_map)
        self.mlp = MLP_Block(input_dim=feature_map.sum_emb_out_dim(), output_dim=1,
                             hidden_units=hidden_units, hidden_activations=hidden_activations,
                             output_activation=None, dropout_rates=net_dropout,
                             batch_norm=batch_norm)
        self.compile(kwargs["optimizer"], kwargs["loss"], learning_rate)
        self.reset_parameters()
        self.model_to_device()

    def forward(self, inputs):
        X = self.get_inputs(inputs)
        feature_emb = self.embedding_layer(X)
        y_pred = self.fm(X, feature_emb)
        y_pred += self.mlp(feature_emb.flatten(start_dim=1))
        y_pred = self.output_activation(y_pred)
        return {"y_pred": y_pred}


class DCNv2(BaseModel):
    def __init__(self, feature_map, model_id="DCNv2", gpu=-1, model_structure="parallel",
                 use_low_rank_mixture=False, low_rank=32, num_experts=4, learning_rate=1e-3,
                 embedding_dim=10, stacked_dnn_hidden_units=[], parallel_dnn_hidden_units=[],
                 dnn_activations="ReLU", num_cross_layers=3, net_dropout=0, batch_norm=False,
                 embedding_regularizer=None, net_regularizer=None, **kwargs):
        super(DCNv2, self).__init__(feature_map, model_id=model_id, gpu=gpu,
                                    embedding_regularizer=embedding_regularizer,
                                    net_regularizer=net_regularizer, **kwargs)
        self.embedding_layer = FeatureEmbedding(feature_map, embedding_dim)
        input_dim = feature_map.sum_emb_out_dim()
        if use_low_rank_mixture:
            raise NotImplementedError("CrossNetMix is outside the hot-path scope (SURVEY §2 #5)")
        self.crossnet = CrossNetV2(input_dim, num_cross_layers)
        self.model_structure = model_structure
        assert self.model_structure in ["crossnet_only", "stacked", "parallel", "stacked_parallel"], \
            "model_structure={} not supported!".format(self.model_structure)
        if self.model_structure in ["stacked", "stacked_parallel"]:
            self.stacked_dnn = MLP_Block(input_dim=input_dim, output_dim=None,
                                         hidden_units=stacked_dnn_hidden_units,
                                         hidden_activations=dnn_activations,
                                         output_activation=None, dropout_rates=net_dropout,
                                         batch_norm=batch_norm)
            final_dim = stacked_dnn_hidden_units[-1]
        if self.model_structure in ["parallel", "stacked_parallel"]:
            self.parallel_dnn = MLP_Block(input_dim=input_dim, output_dim=None,
                                          hidden_units=parallel_dnn_hidden_units,
                                          hidden_activations=dnn_activations,
                                          output_activation=None, dropout_rates=net_dropout,
                                          batch_norm=batch_norm)
            final_dim = input_dim + parallel_dnn_hidden_units[-1]
        if self.model_structure == "stacked_parallel":
            final_dim = stacked_dnn_hidden_units[-1] + parallel_dnn_hidden_units[-1]
        if self.model_structure == "crossnet_only":
            final_dim = input_dim
        self.fc = FxLinear(final_dim, 1, device=self.device)
        self.compile(kwargs["optimizer"], kwargs["loss"], learning_rate)
        self.reset_parameters()
        self.model_to_device()

    def forward(self, inputs):
        X = self.get_inputs(inputs)
        feature_emb = self.embedding_layer(X, flatten_emb=True)
        cross_out = self.crossnet(feature_emb)
        if self.model_structure == "crossnet_only":
            final_out = cross_out
        elif self.model_structure == "stacked":
            final_out = self.stacked_dnn(cross_out)
        elif self.model_structure == "parallel":
            dnn_out = self.parallel_dnn(feature_emb)
            final_out = torch.cat([cross_out, dnn_out], dim=-1)
        elif self.model_structure == "stacked_parallel":
            final_out = torch.cat([self.stacked_dnn(cross_out), self.parallel_dnn(feature_emb)],
                                  dim=-1)
        y_pred = self.fc(final_out)
        y_pred = self.output_activation(y_pred)
        return {"y_pred": y_pred}


def _flatten(items):
    for x in items:
        if isinstance(x, (list, tuple)):
            for y in _flatten(x):
                yield y
        else:
            yield x


class DIN(BaseModel):
    """model_zoo/DIN/src/DIN.py:50-150."""

    def __init__(self, feature_map, model_id="DIN", gpu=-1, dnn_hidden_units=[512, 128, 64],
                 dnn_activations="ReLU", attention_hidden_units=[64],
                 attention_hidden_activations="Dice", attention_output_activation=None,
                 attention_dropout=0, learning_rate=1e-3, embedding_dim=10, net_dropout=0,
                 batch_norm=False, din_target_field=[("item_id", "cate_id")],
                 din_sequence_field=[("click_history", "cate_history")], din_use_softmax=False,
                 embedding_regularizer=None, net_regularizer=None, **kwargs):
        super(DIN, self).__init__(feature_map, model_id=model_id, gpu=gpu,
                                  embedding_regularizer=embedding_regularizer,
                                  net_regularizer=net_regularizer, **kwargs)
        if not isinstance(din_target_field, list):
            din_target_field = [din_target_field]
        self.din_target_field = [tuple(f) if isinstance(f, list) else f for f in din_target_field]
        if not isinstance(din_sequence_field, list):
            din_sequence_field = [din_sequence_field]
        self.din_sequence_field = [tuple(f) if isinstance(f, list) else f
                                   for f in din_sequence_field]
        assert len(self.din_target_field) == len(self.din_sequence_field), \
            "len(din_target_field) != len(din_sequence_field)"
        if isinstance(dnn_activations, str) and dnn_activations.lower() == "dice":
            dnn_activations = [Dice(units) for units in dnn_hidden_units]
        self.feature_map = feature_map
        self.embedding_dim = embedding_dim
        self.embedding_layer = FeatureEmbeddingDict(feature_map, embedding_dim)
        self.attention_layers = nn.ModuleList(
            [DIN_Attention(embedding_dim * len(target_field) if type(target_field) == tuple
                           else embedding_dim,
                           attention_units=attention_hidden_units,
                           hidden_activations=attention_hidden_activations,
                           output_activation=attention_output_activation,
                           dropout_rate=attention_dropout, use_softmax=din_use_softmax)
             for target_field in self.din_target_field])
        self.dnn = MLP_Block(input_dim=feature_map.sum_emb_out_dim(), output_dim=1,
                             hidden_units=dnn_hidden_units, hidden_activations=dnn_activations,
                             output_activation=self.output_activation, dropout_rates=net_dropout,
                             batch_norm=batch_norm)
        self.compile(kwargs["optimizer"], kwargs["loss"], learning_rate)
        self.reset_parameters()
        self.model_to_device()

    def forward(self, inputs):
        X = self.get_inputs(inputs)
        feature_emb_dict = self.embedding_layer(X)
        for idx, (target_field, sequence_field) in enumerate(zip(self.din_target_field,
                                                                 self.din_sequence_field)):
            target_emb = self.get_embedding(target_field, feature_emb_dict)
            sequence_emb = self.get_embedding(sequence_field, feature_emb_dict)
            seq_field = list(_flatten([sequence_field]))[0]
            mask = X[seq_field].long() != 0   # padding_idx = 0 required
            pooling_emb = self.attention_layers[idx](target_emb, sequence_emb, mask)
            for field, field_emb in zip(list(_flatten([sequence_field])),
                                        pooling_emb.split(self.embedding_dim, dim=-1)):
                feature_emb_dict[field] = field_emb
        feature_emb = self.embedding_layer.dict2tensor(feature_emb_dict, flatten_emb=True)
        y_pred = self.dnn(feature_emb)
        return {"y_pred": y_pred}

    def get_embedding(self, field, feature_emb_dict):
        if type(field) == tuple:
            return torch.cat([feature_emb_dict[f] for f in field], dim=-1)
        return feature_emb_dict[field]


class DLRM(BaseModel):
    """model_zoo/DLRM/src/DLRM.py:44-124."""

    def __init__(self, feature_map, model_id="DLRM", gpu=-1, learning_rate=1e-3, embedding_dim=10,
                 top_mlp_units=[64, 64, 64], bottom_mlp_units=[64, 64, 64],
                 top_mlp_activations="ReLU", bottom_mlp_activations="ReLU", top_mlp_dropout=0,
                 bottom_mlp_dropout=0, interaction_op="dot", batch_norm=False,
                 embedding_regularizer=None, net_regularizer=None, **kwargs):
        super(DLRM, self).__init__(feature_map, model_id=model_id, gpu=gpu,
                                   embedding_regularizer=embedding_regularizer,
                                   net_regularizer=net_regularizer, **kwargs)
        self.dense_feats = [feat for feat, spec in feature_map.features.items()
                            if spec["type"] == "numeric"]
        self.embedding_layer = FeatureEmbedding(feature_map, embedding_dim,
                                                not_required_feature_columns=self.dense_feats)
        if len(self.dense_feats) > 0:
            n_fields = feature_map.num_fields - len(self.dense_feats) + 1
            self.bottom_mlp = MLP_Block(input_dim=len(self.dense_feats), output_dim=embedding_dim,
                                        hidden_units=bottom_mlp_units,
                                        hidden_activations=bottom_mlp_activations,
                                        output_activation=bottom_mlp_activations,
                                        dropout_rates=bottom_mlp_dropout, batch_norm=batch_norm)
        else:
            n_fields = feature_map.num_fields
        self.interaction_op = interaction_op
        if self.interaction_op == "dot":
            self.interact = InnerProductInteraction(num_fields=n_fields, output="inner_product")
            top_input_dim = (n_fields * (n_fields - 1)) // 2 + \
                embedding_dim * int(len(self.dense_feats) > 0)
        elif self.interaction_op == "cat":
            self.interact = nn.Flatten(start_dim=1)
            top_input_dim = n_fields * embedding_dim
        else:
            raise ValueError("interaction_op={} not supported.".format(self.interaction_op))
        self.top_mlp = MLP_Block(input_dim=top_input_dim, output_dim=1, hidden_units=top_mlp_units,
                                 hidden_activations=top_mlp_activations,
                                 output_activation=self.output_activation,
                                 dropout_rates=top_mlp_dropout, batch_norm=batch_norm)
        self.compile(kwargs["optimizer"], kwargs["loss"], learning_rate)
        self.reset_parameters()
        self.model_to_device()

    def forward(self, inputs):
        X = self.get_inputs(inputs)
        feat_emb = self.embedding_layer(X)
        if len(self.dense_feats) > 0:
            dense_x = torch.cat([X[k].float().view(-1, 1) for k in self.dense_feats], dim=-1)
            dense_emb = self.bottom_mlp(dense_x)
            feat_emb = torch.cat([feat_emb, dense_emb.unsqueeze(1)], dim=1)
        interact_out = self.interact(feat_emb)
        if self.interaction_op == "dot" and len(self.dense_feats) > 0:
            interact_out = torch.cat([interact_out, dense_emb], dim=-1)
        y_pred = self.top_mlp(interact_out)
        return {"y_pred": y_pred}


class xDeepFM(BaseModel):
    """model_zoo/xDeepFM/src/xDeepFM.py:41-97."""

    def __init__(self, feature_map, model_id="xDeepFM", gpu=-1, learning_rate=1e-3,
                 embedding_dim=10, dnn_hidden_units=[64, 64, 64], dnn_activations="ReLU",
                 cin_hidden_units=[16, 16, 16], net_dropout=0, batch_norm=False,
                 embedding_regularizer=None, net_regularizer=None, **kwargs):
        super(xDeepFM, self).__init__(feature_map, model_id=model_id, gpu=gpu,
                                      embedding_regularizer=embedding_regularizer,
                                      net_regularizer=net_regularizer, **kwargs)
        self.embedding_layer = FeatureEmbedding(feature_map, embedding_dim)
        self.dnn = MLP_Block(input_dim=feature_map.sum_emb_out_dim(), output_dim=1,
                             hidden_units=dnn_hidden_units, hidden_activations=dnn_activations,
                             output_activation=None, dropout_rates=net_dropout,
                             batch_norm=batch_norm) if dnn_hidden_units else None
        self.lr_layer = LogisticRegression(feature_map, use_bias=False)
        self.cin = CompressedInteractionNet(feature_map.num_fields, cin_hidden_units, output_dim=1)
        self.compile(kwargs["optimizer"], kwargs["loss"], learning_rate)
        self.reset_parameters()
        self.model_to_device()

    def forward(self, inputs):
        X = self.get_inputs(inputs)
        feature_emb = self.embedding_layer(X)
        lr_logit = self.lr_layer(X)
        cin_logit = self.cin(feature_emb)
        y_pred = lr_logit + cin_logit
        if self.dnn is not None:
            y_pred = y_pred + self.dnn(feature_emb.flatten(start_dim=1))
        y_pred = self.output_activation(y_pred)
        return {"y_pred": y_pred}
