"""Pin the oracle (oracle/ctr_oracle.py) against golden vectors produced by the REAL reference
(tests/golden/make_golden.py).  Runs on CPU; no GPU, no /root/reference needed."""
import numpy as np
import torch

from oracle import ctr_oracle as O


def _torch_batch(b):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in b.items()}


def test_oracle_forward_matches_reference(golden):
    state = {k: torch.from_numpy(v) for k, v in golden.state0.items()}
    X = _torch_batch(golden.batches[-1])
    with torch.no_grad():
        logit = O.model_logit(golden.cfg(), state, golden.features, X).reshape(-1).numpy()
    np.testing.assert_allclose(logit, golden.expect["logit0"], rtol=0, atol=2e-6)
    pred = O.predict(golden.cfg(), state, golden.features, X).reshape(-1).numpy()
    np.testing.assert_allclose(pred, golden.expect["pred0"], rtol=0, atol=1e-6)


def test_oracle_training_trajectory_matches_reference(golden):
    m = golden.meta
    tr = O.OracleTrainer(golden.cfg(), golden.state0, golden.features, lr=m["lr"],
                         max_norm=m["max_norm"], optimizer=m["optimizer"].lower(),
                         emb_reg=O.parse_regularizer(m.get("emb_reg", 0)),
                         net_reg=O.parse_regularizer(m.get("net_reg", 0)))
    losses = []
    for i in range(m["steps"]):
        b = _torch_batch(golden.batches[i])
        loss, _ = tr.train_step(b, b["label"])
        losses.append(loss)
    np.testing.assert_allclose(losses, golden.expect["loss"], rtol=0, atol=2e-6)
    pred = tr.predict(_torch_batch(golden.batches[-1])).reshape(-1).numpy()
    np.testing.assert_allclose(pred, golden.expect["pred1"], rtol=0, atol=5e-6)
    for k, ref in golden.state1.items():
        got = tr.state[k].detach().numpy()
        np.testing.assert_allclose(got, ref, rtol=0, atol=5e-6, err_msg=k)
