import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from fuxictr_amd import synthetic, zoo
import fuxictr_amd.rank_model as rm
fmap, spec = synthetic.criteo_feature_map(embedding_dim=16)
model = zoo.DeepFM(fmap, model_id="p", gpu=0, embedding_dim=16, hidden_units=[1024] * 4, learning_rate=1e-3,
                   optimizer="adam", loss="binary_crossentropy", task="binary_classification",
                   metrics=["logloss", "AUC"], verbose=0, model_root="/tmp/fxp", hip_graph=True)
model.train()
rng = np.random.default_rng(0)
pool = []
for _ in range(4):
    b = synthetic.criteo_batch(rng, 4096)
    pool.append({k: torch.from_numpy(v.astype(np.float64) if v.dtype == np.float32 else v) for k, v in b.items()})
T = {}
orig_sync = torch.cuda.Event.synchronize
def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter(); r = fn(*a, **k); T[name] = T.get(name, 0.0) + time.perf_counter() - t0; return r
    return w
torch.cuda.Event.synchronize = timed("ev.sync", orig_sync)
model._stage_host_columns = timed("stage", model._stage_host_columns)
for i in range(10):
    model.train_step(pool[i % 4])
torch.cuda.synchronize()
st = model._graph_state
st.fill = timed("fill", st.fill)
st.graph.replay = timed("replay", st.graph.replay)
T.clear()
t0 = time.perf_counter()
N = 100
for i in range(N):
    model.train_step(pool[i % 4])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host loop %.3f ms/step, +drain %.3f ms total" % (1e3 * (t1 - t0) / N, 1e3 * (t2 - t1)))
for k, v in T.items():
    print("  %-8s %.3f ms/step" % (k, 1e3 * v / N))
