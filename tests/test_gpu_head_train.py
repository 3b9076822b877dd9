"""fx_head_train (round 4): the Linear(K -> 1) head forward + sigmoid / BCE + the head's backward in one
pass over the top hidden layer, against the three kernels it replaces (fx_gemm_f32 N = 1, fx_sigmoid_bce,
the head backward of fx_gemm_f32_batch): logit, dlogit and the input gradient bit for bit (same
expressions, same order), dW / db / loss — sums over the batch in another fixed order — against float64."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,K,mask,use_add,use_bias,scale", [
    (4096, 1024, True, True, True, 1.0),        # DeepFM: FM + first-order term rides in `add`
    (4096, 1648, False, False, True, 1.0),      # DCNv2's fc over [cross | deep]
    (4096, 64, True, False, True, 0.125),       # DIN's tower, one rank of eight
    (1000, 256, True, True, False, 1.0),
    (37, 12, False, True, True, 0.5),
    (3, 2048, True, False, True, 1.0),
])
def test_head_train_equals_the_three_kernels(M, K, mask, use_add, use_bias, scale):
    from fuxictr_amd import ops
    from fuxictr_amd.layers import linear_grads
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(M + K)
    h = torch.randn(M, K, generator=g).to(dev)
    if mask:
        h = torch.relu(h)
    W = (torch.randn(1, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(1, generator=g).to(dev) if use_bias else None
    add = torch.randn(M, 1, generator=g).to(dev) if use_add else None
    y = (torch.rand(M, 1, generator=g) > 0.7).float().to(dev)
    # the separate kernels
    logit0 = torch.empty(M, 1, device=dev)
    ops.gemm(h, W, logit0, transb=True, bias=b, add=add)
    loss0, dl0 = torch.empty((), device=dev), torch.empty(M, 1, device=dev)
    ops.sigmoid_bce(logit0, y, loss=loss0, dlogit=dl0)
    if scale != 1.0:
        dl0 = dl0 * scale
    dW0, db0, dz0 = linear_grads(dl0, h, W, use_bias, mask=h if mask else None)
    # the one pass
    logit, dl = torch.empty(M, 1, device=dev), torch.empty(M, 1, device=dev)
    dz, dW = torch.empty(M, K, device=dev), torch.empty(1, K, device=dev)
    db = torch.empty(1, device=dev) if use_bias else None
    loss = torch.empty((), device=dev)
    ws = torch.empty(ops.head_train_workspace_floats(M, K), device=dev)
    assert ops.head_train_ok(h, W, add)
    ops.head_train(h, W, b, add, y, 0 if mask else -1, scale, logit, dl, dz, dW, db, loss, ws)
    torch.cuda.synchronize()
    assert torch.equal(logit, logit0)
    assert torch.equal(dl, dl0)
    assert torch.equal(dz, dz0)
    # sums over the batch: float64 reference
    d64 = dl.double().cpu()
    dW64 = (d64 * h.double().cpu()).sum(0, keepdim=True)
    tol = 4e-6 * max(1.0, float(dW64.abs().max())) * max(1.0, (M / 4096.0) ** 0.5)
    assert float((dW.double().cpu() - dW64).abs().max()) <= tol
    assert float((dW0.double().cpu() - dW64).abs().max()) <= tol
    if use_bias:
        assert abs(float(db.item()) - float(d64.sum())) <= 1e-6
        assert abs(float(db0.reshape(-1)[0].item()) - float(d64.sum())) <= 1e-6
    p64 = torch.sigmoid(logit.double().cpu())
    l64 = torch.nn.functional.binary_cross_entropy(p64, y.double().cpu())
    assert abs(float(loss.item()) - float(l64)) <= 2e-6 * max(1.0, float(l64))
    assert abs(float(loss0.item()) - float(l64)) <= 2e-6 * max(1.0, float(l64))


def test_head_train_without_input_gradient():
    from fuxictr_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    M, K = 512, 128
    h, W = torch.randn(M, K, generator=g).to(dev), torch.randn(1, K, generator=g).to(dev)
    y = (torch.rand(M, 1, generator=g) > 0.5).float().to(dev)
    logit, dl, dW = torch.empty(M, 1, device=dev), torch.empty(M, 1, device=dev), torch.empty(1, K, device=dev)
    loss = torch.empty((), device=dev)
    ws = torch.empty(ops.head_train_workspace_floats(M, K), device=dev)
    ops.head_train(h, W, None, None, y, -1, 1.0, logit, dl, None, dW, None, loss, ws)
    torch.cuda.synchronize()
    ref = (dl.double() * h.double()).sum(0, keepdim=True)
    assert float((dW.double() - ref).abs().max()) <= 1e-5
    assert np.isfinite(float(loss.item()))


def test_head_train_masks_from_a_column_on():
    """DCNv2's head reads [cross | deep]: only the deep part went through a ReLU (mask_from = 624)."""
    from fuxictr_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(2)
    M, K, D0 = 2048, 1648, 624
    h = torch.randn(M, K, generator=g).to(dev)
    h[:, D0:] = torch.relu(h[:, D0:])
    W = (torch.randn(1, K, generator=g) / K ** 0.5).to(dev)
    y = (torch.rand(M, 1, generator=g) > 0.5).float().to(dev)
    outs = []
    for mf in (-1, D0):
        logit, dl = torch.empty(M, 1, device=dev), torch.empty(M, 1, device=dev)
        dz, dW = torch.empty(M, K, device=dev), torch.empty(1, K, device=dev)
        loss = torch.empty((), device=dev)
        ws = torch.empty(ops.head_train_workspace_floats(M, K), device=dev)
        ops.head_train(h, W, None, None, y, mf, 1.0, logit, dl, dz, dW, None, loss, ws)
        outs.append((dz.clone(), dW.clone(), dl.clone()))
    torch.cuda.synchronize()
    (dz0, dW0, dl0), (dz1, dW1, dl1) = outs
    assert torch.equal(dl0, dl1) and torch.equal(dW0, dW1)
    assert torch.equal(dz1[:, :D0], dz0[:, :D0])
    assert torch.equal(dz1[:, D0:], torch.where(h[:, D0:] > 0, dz0[:, D0:], torch.zeros((), device=dev)))
