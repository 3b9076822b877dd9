"""A/B of SURVEY.md 8f-3 on one MI355X: DeepFM on the synthetic Taobao-shape data (14 categorical
fields + a 50-long click history behind MaskedAveragePooling — the default encoder of a sequence
feature), full training step at B=4096 replayed from a hipGraph, with the pooling fused into the
gather (one slot per history) and with the unfused layout ([B, 50, 16] history written by the
gather, reduced by torch ops, features re-stacked).  Prints one JSON line."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/scripts/", 1)[0])
from fuxictr_amd import layers, synthetic, zoo  # noqa: E402
from fuxictr_amd.features import FeatureMap  # noqa: E402


def run(fused, B=4096, steps=50, warmup=8):
    layers.FeatureEmbeddingDict.fuse_pooling = fused
    _, spec = synthetic.taobao_feature_map(embedding_dim=16)
    for item in spec["features"]:
        (name, fs), = item.items()
        if fs["type"] == "sequence":
            fs["feature_encoder"] = "layers.MaskedAveragePooling()"
    fmap = FeatureMap(spec["dataset_id"], data_dir="")
    fmap.load_dict(spec, {"embedding_dim": 16})
    torch.manual_seed(2019)
    model = zoo.DeepFM(fmap, model_id="seqpool", hidden_units=[1024] * 4, gpu=0, embedding_dim=16,
                       learning_rate=1e-3, optimizer="adam", loss="binary_crossentropy",
                       task="binary_classification", metrics=["logloss", "AUC"], verbose=0,
                       model_root="/tmp/fx_bench", hip_graph=True)
    rng = np.random.default_rng(0)
    batches = []
    for _ in range(4):
        b = synthetic.taobao_batch(rng, B, spec)
        batches.append({k: torch.from_numpy(v).to("cuda:0") for k, v in b.items()})
    model.train()
    losses = []
    for i in range(warmup):
        losses.append(float(model.train_step(batches[i % 4]).detach()))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        model.train_step(batches[i % 4])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    return ms, losses


if __name__ == "__main__":
    ms_f, l_f = run(True)
    ms_u, l_u = run(False)
    L, D = 50, 16
    print(json.dumps({
        "workload": "DeepFM, synthetic Taobao-shape (14 fields + click history L=50, mean pooled), "
                    "B=4096, Adam, hipGraph replay",
        "fused_ms_per_step": ms_f, "unfused_ms_per_step": ms_u,
        "fused_samples_per_sec": 4096e3 / ms_f, "unfused_samples_per_sec": 4096e3 / ms_u,
        "history_bytes_not_written_per_step": 4096 * L * D * 4,
        "first_losses_fused": l_f[:3], "first_losses_unfused": l_u[:3]}))
