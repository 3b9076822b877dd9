"""TEST INFRASTRUCTURE: torch-CPU emulation of the libfxctr entry points, with the same
signatures as fuxictr_amd.ops.  tests/test_host_wiring.py monkeypatches it in so the HOST logic
(layer bookkeeping, autograd wiring, optimizer sequencing, BaseModel loop) can be exercised in the
GPU-less build container.  It is never importable from the product package.
It is a SECOND implementation of every kernel's contract: the CPU tests that run on it validate the host
wiring, not the HIP code — the kernels themselves are held to the oracle / numpy / fp64 restatements by the
`-m gpu` tests, which call through the C-ABI."""
import math

import numpy as np
import torch

from fuxictr_amd import _lib

SC = _lib


def new_scalars(device, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, max_norm=0.0, series=False):
    s = torch.zeros(_lib.SC_WORDS, dtype=torch.float32)
    s[SC.SC_LR], s[SC.SC_BETA1], s[SC.SC_BETA2], s[SC.SC_EPS] = lr, beta1, beta2, eps
    s[SC.SC_CLIP], s[SC.SC_MAX_NORM] = 1.0, max_norm
    return s


def _step(scal):
    return int(scal.view(torch.int32)[SC.SC_STEP])


def pack_columns(cols, out, out_col0=0):
    c = out_col0
    for t in cols:
        t2 = t.reshape(t.shape[0], -1)
        out[:, c:c + t2.shape[1]] = t2.to(out.dtype)
        c += t2.shape[1]
    return out


def emb_gather_fwd(table, D, ids, col_row_base, col_vocab, col_out_off, dense, num_w,
                   num_out_off, out, scal, n_cols=None):
    if ids is not None:
        for c in range(ids.shape[1] if n_cols is None else n_cols):
            rows = ids[:, c].long() + int(col_row_base[c])
            o = int(col_out_off[c])
            out[:, o:o + D] = table[rows]
    if dense is not None:
        for j in range(dense.shape[1]):
            o = int(num_out_off[j])
            out[:, o:o + D] = dense[:, j:j + 1] * num_w[j]
    return out


POOL_SUM, POOL_MEAN = 0, 1


def emb_seq_pool_fwd(table, D, ids, col_row_base, col_vocab, seq_col0, seq_len, seq_mode,
                     seq_out_off, out, denom, scal):
    for s in range(denom.shape[1]):
        c0, L = int(seq_col0[s]), int(seq_len[s])
        rows = table[ids[:, c0:c0 + L].long() + col_row_base[c0:c0 + L].view(1, -1)]   # [B, L, D]
        den = (rows.sum(-1) != 0).float().sum(-1, keepdim=True) + 1e-12
        pooled = rows.sum(1)
        if int(seq_mode[s]) == POOL_MEAN:
            pooled = pooled / den
        o = int(seq_out_off[s])
        out[:, o:o + D] = pooled
        denom[:, s:s + 1] = den
    return out


def dedup_workspace_bytes(n):
    return 256


class DedupResult(object):
    pass


def dedup(ids, col_row_base, col_vocab, col_pad, total_rows, workspace, result=None,
          n_shards=1, want_uid=False, columns_sorted=False, begin_scal=None, grouped=False):
    # (grouped: the caller accepts any deterministic grouping of the rows; ascending is one)
    if begin_scal is not None:
        opt_begin_step(begin_scal)
    B, C = ids.shape
    keys = ids.long() + col_row_base.view(1, -1)
    valid = (ids != col_pad.view(1, -1)) & (ids >= 0) & (ids < col_vocab.view(1, -1))
    sentinel = total_rows
    if n_shards > 1:
        rps = -(-total_rows // n_shards)
        keys = (keys % n_shards) * rps + keys // n_shards
        sentinel = rps * n_shards
    keys = torch.where(valid, keys, torch.full_like(keys, sentinel)).reshape(-1)
    skey, spos = torch.sort(keys, stable=True)
    nvalid = int(valid.sum())
    uniq, counts = torch.unique_consecutive(skey[:nvalid], return_counts=True)
    dd = DedupResult()
    n = B * C
    dd.sorted_key, dd.sorted_pos = skey.int(), spos.int()
    dd.uniq_row = torch.zeros(n, dtype=torch.int32)
    dd.uniq_row[:len(uniq)] = uniq.int()
    dd.seg_start = torch.zeros(n + 1, dtype=torch.int32)
    dd.seg_start[1:len(uniq) + 1] = torch.cumsum(counts, 0).int()
    dd.n_unique = torch.tensor([len(uniq)], dtype=torch.int32)
    dd.n_max, dd.C = n, C
    dd.sorted_uid = None
    if want_uid:
        uid = torch.full((n,), -1, dtype=torch.int32)
        if nvalid:
            uid[:nvalid] = (torch.repeat_interleave(torch.arange(len(uniq)), counts)).int()
        dd.sorted_uid = uid
    return dd


def shard_plan_workspace_ints(n_lookups, n_shards):
    return 8


def shard_plan(dd, n_shards, total_rows, cap, send_idx, uniq_slot, lookup_slot, scal,
               global_keys=False, workspace=None, slot_uniq=None):
    rps = -(-total_rows // n_shards)
    nu = int(dd.n_unique)
    keys = dd.uniq_row[:nu].long()
    send_idx.fill_(rps)
    uniq_slot.fill_(n_shards * cap)
    if slot_uniq is not None:
        slot_uniq.fill_(-1)
    for o in range(n_shards):
        if global_keys:      # keys are global rows: owner = g % N, local row = g // N
            sel = (keys % n_shards == o).nonzero().reshape(-1)
            local = keys[sel] // n_shards
        else:
            sel = ((keys >= o * rps) & (keys < (o + 1) * rps)).nonzero().reshape(-1)
            local = keys[sel] - o * rps
        if len(sel) > cap:
            scal.view(torch.int32)[SC.SC_ERR] |= _lib.FX_FLAG_A2A_OVERFLOW
            sel, local = sel[:cap], local[:cap]
        send_idx[o * cap:o * cap + len(sel)] = local.int()
        uniq_slot[sel] = (o * cap + torch.arange(len(sel))).int()
        if slot_uniq is not None:
            slot_uniq[o * cap:o * cap + len(sel)] = sel.int()
    flat = lookup_slot.view(-1)
    uid = dd.sorted_uid.long()
    slots = torch.where(uid >= 0, uniq_slot[uid.clamp(min=0)].long(),
                        torch.full_like(uid, n_shards * cap))
    flat[dd.sorted_pos.long()] = slots.int()


def fill_grad_block(tables, slot_uniq, block):
    block.zero_()
    on = (slot_uniq >= 0).nonzero().reshape(-1)
    for G, D, off in tables:
        if G is not None:
            block[on, off:off + D] = G[slot_uniq[on].long()]


def owner_grad_reduce_partials(n_max):
    return max(1, -(-n_max // 32))


def owner_grad_reduce(grecv, dd, tables, sq_partials):
    nu = int(dd.n_unique)
    sq_partials.zero_()
    seg = dd.seg_start.long()
    pos = dd.sorted_pos.long()
    for G, D, off in tables:
        if G is None:
            continue
        for u in range(nu):
            acc = torch.zeros(D)
            for i in range(int(seg[u]), int(seg[u + 1])):      # ascending rank order, like the kernel
                acc = acc + grecv[pos[i], off:off + D]
            G[u] = acc
            sq_partials[u // 32] += float((acc * acc).sum())


def owner_fetch_rows(states, offs, dd, send, catchup, scal, upto_offset=-1, zero_row=None):
    if zero_row is not None:
        zero_row.zero_()
    send.zero_()
    nu = int(dd.n_unique)
    seg = dd.seg_start.long()
    pos = dd.sorted_pos.long()
    rows = dd.uniq_row[:nu].long()
    # position -> unique row index of its run
    owner = torch.repeat_interleave(torch.arange(nu), (seg[1:nu + 1] - seg[:nu]))
    for st, off in zip(states, offs):
        if catchup:
            adam_catchup(st.table, st.m, st.v, st.last_step, st.D, dd, st.table.shape[0],
                         upto_offset, scal)
        if nu:
            send[pos[:int(seg[nu])], off:off + st.D] = st.table[rows[owner]].float()


def scatter_rows(src, row_map, n_rows, n_max, D, dst):
    n = int(n_rows)
    dst[row_map[:n].long()] = src[:n]


def split_rows(src, n_rows, parts, zero_tail_rows=0):
    for off, dst in parts:
        dst[:n_rows] = src[:n_rows, off:off + dst.shape[1]]
        dst[n_rows:n_rows + zero_tail_rows] = 0


def sum_parts(parts, out):
    out[0] = sum(float(p.double().sum()) for p in parts)


def emb_grad_reduce_partials(n_max, D):
    return max(1, (n_max + 3) // 4)


def emb_grad_reduce_scratch_ints(n_max):
    return n_max // 33 + 2


def _flat_from(t):
    """What a kernel sees: the memory from t's first element on (t may be a strided view)."""
    n = t.untyped_storage().nbytes() // t.element_size() - t.storage_offset()
    return torch.as_strided(t, (n,), (1,), t.storage_offset())


def emb_grad_reduce(dout, dout_ld, col_out_off, C, D, dd, G, sq_partials, scratch,
                    col_denom=None, denom=None):
    nu = int(dd.n_unique)
    flat = _flat_from(dout)
    sq_partials.zero_()
    for u in range(nu):
        acc = torch.zeros(D)
        for i in range(int(dd.seg_start[u]), int(dd.seg_start[u + 1])):
            p = int(dd.sorted_pos[i])
            if p == 0xFFFFFFFF:          # padding_idx / bad id of the column fast path
                continue
            b, c = divmod(p, C)
            o = b * dout_ld + int(col_out_off[c])
            if col_denom is not None and int(col_denom[c]) >= 0:
                acc += flat[o:o + D] / denom[b, int(col_denom[c])]
            else:
                acc += flat[o:o + D]
        G[u] = acc
    sq_partials[0] = float((G[:nu].double() ** 2).sum())


def emb_numeric_grad(dout, dout_ld, num_out_off, dense, D, dnum_w):
    B = dense.shape[0]
    flat = dout.reshape(-1)
    for j in range(dense.shape[1]):
        o = int(num_out_off[j])
        idx = (torch.arange(B) * dout_ld + o).view(-1, 1) + torch.arange(D).view(1, -1)
        dnum_w[j] = (dense[:, j:j + 1] * flat[idx]).sum(0)


def _beta_f64(b32):
    """fx_dec_f64 (csrc/fx_common.h): the python-side double behind the fp32 image of lr / a beta."""
    x = float(b32)
    if not (0.0 < x < 1e30):
        return x
    r = float("%.6g" % x)
    return r if float(torch.tensor(r, dtype=torch.float32)) == x else x


def opt_begin_step(scal):
    t = _step(scal) + 1
    scal.view(torch.int32)[SC.SC_STEP] = t
    b1, b2 = _beta_f64(scal[SC.SC_BETA1]), _beta_f64(scal[SC.SC_BETA2])
    bc1 = 1 - b1 ** t
    scal[SC.SC_BC1] = bc1
    scal[SC.SC_BC2S] = math.sqrt(1 - b2 ** t)
    scal[SC.SC_STEP_SIZE] = _beta_f64(scal[SC.SC_LR]) / bc1


def clip_coef(parts, scal):
    total = math.sqrt(sum(float(p.double().sum()) for p in parts))
    mx = float(scal[SC.SC_MAX_NORM])
    scal[SC.SC_TOTAL_NORM] = total
    scal[SC.SC_CLIP] = min(1.0, mx / (total + 1e-6)) if mx > 0 else 1.0


def _adam(p, m, v, g, scal):
    b1, b2, eps = scal[SC.SC_BETA1], scal[SC.SC_BETA2], scal[SC.SC_EPS]
    w1 = float(torch.tensor(1.0 - _beta_f64(b1), dtype=torch.float32))     # torch's float(1 - beta)
    w2 = float(torch.tensor(1.0 - _beta_f64(b2), dtype=torch.float32))
    m += w1 * (g - m)
    v.mul_(b2).add_(w2 * g * g)
    p -= scal[SC.SC_STEP_SIZE] * (m / (v.sqrt() / scal[SC.SC_BC2S] + eps))


def _reg(p, scal):
    return scal[SC.SC_REG_L1] * torch.sign(p) + scal[SC.SC_REG_L2] * p


def reg_stats(x, scal, partials):
    nb = _lib.FX_REG_BLOCKS
    partials.zero_()
    partials[0] = float((x.double() ** 2).sum())
    partials[nb] = float(x.double().abs().sum())
    partials[2 * nb] = float((_reg(x, scal).double() ** 2).sum())


def reg_cross(table, D, dd, G, scal, partials):
    nu = int(dd.n_unique)
    rows = dd.uniq_row[:nu].long()
    partials.zero_()
    partials[0] = float((2.0 * G[:nu].double() * _reg(table[rows], scal).double()).sum())


def reg_dense_update(table, m, v, last_step, D, adam, scal):
    rows = (last_step != _step(scal)).nonzero().view(-1)
    p = table[rows]
    g = _reg(p, scal) * scal[SC.SC_CLIP]
    if adam:
        mm, vv = m[rows], v[rows]
        _adam(p, mm, vv, g, scal)
        m[rows], v[rows] = mm, vv
    else:
        p -= scal[SC.SC_LR] * g
    table[rows] = p


def sparse_adam(table, m, v, last_step, D, dd, G, scal):
    nu = int(dd.n_unique)
    rows = dd.uniq_row[:nu].long()
    p, mm, vv = table[rows].float(), m[rows], v[rows]          # bf16 tables: fp32 arithmetic,
    _adam(p, mm, vv, (G[:nu] + _reg(p, scal)) * scal[SC.SC_CLIP], scal)
    table[rows], m[rows], v[rows] = p.to(table.dtype), mm, vv  # ... rounded on the way back
    last_step[rows] = _step(scal)


def adam_catchup(table, m, v, last_step, D, dd, total_rows, upto_offset, scal):
    rows = torch.arange(total_rows) if dd is None else dd.uniq_row[:int(dd.n_unique)].long()
    upto = _step(scal) + upto_offset
    b1, b2, eps, lr = (float(scal[SC.SC_BETA1]), float(scal[SC.SC_BETA2]),
                       float(scal[SC.SC_EPS]), float(scal[SC.SC_LR]))
    for r in rows.tolist():
        last = int(last_step[r])
        p = table[r].float()
        for t in range(last + 1, upto + 1):
            m[r] *= b1
            v[r] *= b2
            p -= lr / (1 - b1 ** t) * (m[r] / (v[r].sqrt() / math.sqrt(1 - b2 ** t) + eps))
        if upto > last:
            table[r] = p.to(table.dtype)
            last_step[r] = upto


def sparse_sgd(table, D, dd, G, scal, last_step=None):
    nu = int(dd.n_unique)
    rows = dd.uniq_row[:nu].long()
    table[rows] -= scal[SC.SC_LR] * scal[SC.SC_CLIP] * (G[:nu] + _reg(table[rows], scal))
    if last_step is not None:
        last_step[rows] = _step(scal)


def mt_sqnorm(grads, sq_partials):
    sq_partials.zero_()
    for i, g in enumerate(grads):
        sq_partials[i * _lib.FX_MT_BLOCKS] = float((g.double() ** 2).sum())


def mt_adam(params, grads, ms, vs, scal):
    for p, g, m, v in zip(params, grads, ms, vs):
        _adam(p, m, v, g * scal[SC.SC_CLIP], scal)


def mt_sgd(params, grads, scal):
    for p, g in zip(params, grads):
        p -= scal[SC.SC_LR] * scal[SC.SC_CLIP] * g


def fm_fwd(emb, F, D, addend, out):
    e = emb.view(-1, F, D)
    r = 0.5 * ((e.sum(1) ** 2) - (e ** 2).sum(1)).sum(-1, keepdim=True)
    out.copy_(r + (addend if addend is not None else 0))
    return out


def fm_bwd(emb, F, D, g, demb, accumulate=False):
    e = emb.view(-1, F, D)
    d = g.view(-1, 1, 1) * (e.sum(1, keepdim=True) - e)
    d = d.reshape(demb.shape)
    demb.copy_(demb + d if accumulate else d)
    return demb


def lr_fwd(table1, ids, col_row_base, col_vocab, dense, num_w1, bias, out, scal):
    acc = torch.zeros(out.shape[0], 1)
    if ids is not None:
        for c in range(ids.shape[1]):
            acc += table1[ids[:, c].long() + int(col_row_base[c])]
    if dense is not None:
        acc += dense @ num_w1.reshape(-1, 1)
    out.copy_(acc + (bias if bias is not None else 0))
    return out


def gemm(A, B_, C_, transa=False, transb=False, bias=None, act=0, zout=None, mul=None, mask=None,
         add=None, split_k=1, workspace=None, rowsum=None):
    a = A.t() if transa else A
    b = B_.t() if transb else B_
    if rowsum is not None:
        rowsum.copy_(a.sum(1))
    z = a @ b
    if bias is not None:
        z = z + bias
    if zout is not None:
        zout.copy_(z)
    if act == 1:
        z = z.clamp(min=0)
    if mul is not None:
        z = z * mul
    if mask is not None:
        z = torch.where(mask > 0, z, torch.zeros(()))
    if add is not None:
        z = z + add
    C_.copy_(z)
    return C_


def gemm_problem(A, B_, C_, **kw):
    return (A, B_, C_, kw)


def gemm_batch(problems):
    for A, B_, C_, kw in problems:
        gemm(A, B_, C_, **kw)


def gemm_workspace_floats(M, N, split_k):
    return split_k * M * (N + 1) + 64


def gemm_dw_dx(dz, x, W, dW, dx, split_k=1, workspace=None, rowsum=None, mask=None, add=None):
    gemm(dz, x, dW, transa=True, transb=False, split_k=split_k, workspace=workspace, rowsum=rowsum)
    gemm(dz, W, dx, transa=False, transb=False, mask=mask, add=add)


def colsum(X, out, workspace):
    out.copy_(X.sum(0))
    return out


def mask_mul(dy, y, out):
    out.copy_(torch.where(y > 0, dy, torch.zeros(())))
    return out


def cross_bwd_prep(dxn, x0, z, t, dx0, init, add_dxn):
    t.copy_(dxn * x0)
    term = dxn * z + (dxn if add_dxn else 0)
    dx0.copy_(term if init else dx0 + term)


def sigmoid_bce(logit, y, prob=None, loss=None, dlogit=None):
    p = torch.sigmoid(logit)
    if prob is not None:
        prob.copy_(p)
    if y is None:
        return
    if loss is not None:
        loss.copy_(torch.nn.functional.binary_cross_entropy(p, y))
    if dlogit is not None:
        dlogit.copy_((p - y) / torch.clamp((1 - p) * p, min=1e-12) / logit.numel() * (p * (1 - p)))


def head_train_workspace_floats(M, K):
    return min(256, max(1, -(-M // 4))) * (K + 2)


def head_train_ok(h, W, out_add=None):
    K = W.shape[1]
    return (W.shape[0] == 1 and K % 4 == 0 and 8 < K <= 2048 and h.dim() == 2 and h.stride(1) == 1
            and (out_add is None or (out_add.numel() == h.shape[0] and out_add.dim() <= 2)))


def head_train(h, W, bias, out_add, y, mask_from, root_scale, logit, dlogit, dz, dW, db, loss, workspace):
    M = h.shape[0]
    z = h @ W.reshape(-1)
    if bias is not None:
        z = z + bias.reshape(())
    if out_add is not None:
        z = z + out_add.reshape(-1)
    yv = y.reshape(-1)
    p = torch.sigmoid(z)
    loss.copy_(torch.nn.functional.binary_cross_entropy(p, yv))
    d = (p - yv) / torch.clamp((1 - p) * p, min=1e-12) / M * (p * (1 - p))
    if root_scale != 1.0:
        d = d * root_scale
    logit.reshape(-1).copy_(z)
    dlogit.reshape(-1).copy_(d)
    if dz is not None:
        o = d[:, None] * W.reshape(1, -1)
        if mask_from >= 0:
            o[:, mask_from:] = torch.where(h[:, mask_from:] > 0, o[:, mask_from:], torch.zeros(()))
        dz.copy_(o)
    dW.copy_((d[:, None] * h).sum(0).reshape(dW.shape))
    if db is not None:
        db.copy_(d.sum().reshape(db.shape))


def din_concat_fwd(q, K, out):
    B, L, E = K.shape
    t = q.unsqueeze(1).expand(-1, L, -1)
    out.copy_(torch.cat([t, K, t - K, t * K], dim=-1).reshape(B * L, 4 * E))
    return out


def din_concat_bwd(dx, q, K, dq, dK):
    B, L, E = K.shape
    d = dx.view(B, L, 4, E)
    dq.copy_((d[:, :, 0] + d[:, :, 2] + d[:, :, 3] * K).sum(1))
    dK.copy_(d[:, :, 1] - d[:, :, 2] + d[:, :, 3] * q.unsqueeze(1))


def din_pool_fwd(w, ids, K, out):
    out.copy_(((w * (ids != 0).float()).unsqueeze(-1) * K).sum(1))
    return out


def din_pool_bwd(w, ids, K, dout, dw, dK):
    m = (ids != 0).float()
    dw.copy_(m * (dout.unsqueeze(1) * K).sum(-1))
    dK.copy_((w * m).unsqueeze(-1) * dout.unsqueeze(1))


def dice_workspace_floats(H):
    return 256 * 3 * H


def _dice_stats(Z, training, running_mean, running_var, momentum, stats, update):
    H = Z.shape[1]
    if training:
        mean = Z.double().mean(0)
        var = Z.double().var(0, unbiased=False)
        if update:
            N = Z.shape[0]
            running_mean.mul_(1 - momentum).add_(momentum * mean.float())
            running_var.mul_(1 - momentum).add_(momentum * (var * N / max(N - 1, 1)).float())
        stats[:H] = mean.float()
        stats[H:] = var.float()
    else:
        stats[:H] = running_mean
        stats[H:] = running_var


def dice_fwd(Z, alpha, eps, momentum, training, running_mean, running_var, stats, Y, workspace):
    H = Z.shape[1]
    _dice_stats(Z, training, running_mean, running_var, momentum, stats, True)
    zh = (Z - stats[:H]) / torch.sqrt(stats[H:] + eps)
    p = torch.sigmoid(zh)
    Y.copy_(p * Z + alpha * (1 - p) * Z)
    return Y


def dice_bwd(Z, dY, alpha, eps, training, stats, dZ, dalpha, workspace):
    N, H = Z.shape
    rstd = 1.0 / torch.sqrt(stats[H:] + eps)
    zh = (Z - stats[:H]) * rstd
    p = torch.sigmoid(zh)
    dalpha.copy_((dY * (1 - p) * Z).sum(0))
    dzh = dY * Z * (1 - alpha) * p * (1 - p)
    if training:
        dzh = dzh - dzh.mean(0) - zh * (dzh * zh).mean(0)
    dZ.copy_(dY * (p + alpha * (1 - p)) + dzh * rstd)


def dice_local_sums(Z, sums, workspace):
    H = Z.shape[1]
    sums[:H] = Z.sum(0)
    sums[H:2 * H] = (Z * Z).sum(0)


def dice_fwd_from_sums(Z, alpha, eps, momentum, sums, n_total, running_mean, running_var, stats, Y):
    H = Z.shape[1]
    mean = sums[:H].double() / n_total
    var = (sums[H:2 * H].double() / n_total - mean * mean).clamp(min=0)
    stats[:H] = mean.float()
    stats[H:] = var.float()
    unb = var * n_total / (n_total - 1) if n_total > 1 else var
    running_mean.copy_(((1 - momentum) * running_mean.double() + momentum * mean).float())
    running_var.copy_(((1 - momentum) * running_var.double() + momentum * unb).float())
    zh = (Z - stats[:H]) / torch.sqrt(stats[H:] + eps)
    p = torch.sigmoid(zh)
    Y.copy_(p * Z + alpha * (1 - p) * Z)
    return Y


def dice_bwd_local_sums(Z, dY, alpha, eps, stats, sums3, workspace):
    H = Z.shape[1]
    rstd = 1.0 / torch.sqrt(stats[H:] + eps)
    zh = (Z - stats[:H]) * rstd
    p = torch.sigmoid(zh)
    dzh = dY * Z * (1 - alpha) * p * (1 - p)
    sums3[:H] = (dY * (1 - p) * Z).sum(0)
    sums3[H:2 * H] = dzh.sum(0)
    sums3[2 * H:] = (dzh * zh).sum(0)


def dice_bwd_from_sums(Z, dY, alpha, eps, stats, sums3, n_total, dZ):
    H = Z.shape[1]
    rstd = 1.0 / torch.sqrt(stats[H:] + eps)
    zh = (Z - stats[:H]) * rstd
    p = torch.sigmoid(zh)
    dzh = dY * Z * (1 - alpha) * p * (1 - p)
    dzh = dzh - sums3[H:2 * H] / n_total - zh * (sums3[2 * H:] / n_total)
    dZ.copy_(dY * (p + alpha * (1 - p)) + dzh * rstd)


# ---- DIN attention with the MLP fused in (fx_din_attn.hip) ----------------------------------------
DIN_ATTN_MAX_E, DIN_ATTN_MAX_H = 16, 64


def din_attn_workspace_floats(B, L, E, H):
    return 1


def _din_attn_h(q, K, W1, b1):
    B, L, E = K.shape
    t = q.unsqueeze(1).expand(-1, L, -1)
    x = torch.cat([t, K, t - K, t * K], dim=-1).reshape(B * L, 4 * E)
    h = x @ W1.t()
    if b1 is not None:
        h = h + b1
    return x, h


def din_attn_stats(q, K, W1, b1, sums, workspace, stats=None, momentum=0.0, running_mean=None,
                   running_var=None, num_batches_tracked=None):
    H = W1.shape[0]
    _, h = _din_attn_h(q, K, W1, b1)
    sums[:H] = h.sum(0)
    sums[H:2 * H] = (h * h).sum(0)
    if stats is not None:
        dice_stats_from_sums(sums, H, h.shape[0], momentum, True, running_mean, running_var, stats,
                             num_batches_tracked)


def dice_stats_from_sums(sums, H, n_total, momentum, training, running_mean, running_var, stats,
                         num_batches_tracked=None):
    if training and num_batches_tracked is not None:
        num_batches_tracked += 1
    if not training:
        stats[:H] = running_mean
        stats[H:] = running_var
        return
    mean = sums[:H].double() / n_total
    var = (sums[H:2 * H].double() / n_total - mean * mean).clamp(min=0)
    stats[:H] = mean.float()
    stats[H:] = var.float()
    unb = var * n_total / (n_total - 1) if n_total > 1 else var
    running_mean.copy_(((1 - momentum) * running_mean.double() + momentum * mean).float())
    running_var.copy_(((1 - momentum) * running_var.double() + momentum * unb).float())


def _din_attn_gate(h, alpha, eps, stats):
    H = h.shape[1]
    rstd = 1.0 / torch.sqrt(stats[H:] + eps)
    zh = (h - stats[:H]) * rstd
    p = torch.sigmoid(zh)
    return rstd, zh, p


def _din_attn_mask(mask, B, L):
    return torch.ones(B, L) if mask is None else (mask != 0).float()


def din_attn_fwd(q, K, W1, b1, alpha, eps, stats, W2, b2, mask, a_out, out):
    B, L, E = K.shape
    _, h = _din_attn_h(q, K, W1, b1)
    _, _, p = _din_attn_gate(h, alpha, eps, stats)
    y = p * h + alpha * (1 - p) * h
    a = y @ W2.reshape(-1)
    if b2 is not None:
        a = a + b2
    a_out.copy_(a.view(a_out.shape))
    out.copy_(((a.view(B, L) * _din_attn_mask(mask, B, L)).unsqueeze(-1) * K).sum(1))
    return out


def din_attn_bwd_sums(q, K, W1, b1, alpha, eps, stats, W2, mask, dout, da, sums5, workspace):
    B, L, E = K.shape
    H = W1.shape[0]
    da.copy_((_din_attn_mask(mask, B, L) * (dout.unsqueeze(1) * K).sum(-1)).view(da.shape))
    _, h = _din_attn_h(q, K, W1, b1)
    _, zh, p = _din_attn_gate(h, alpha, eps, stats)
    y = p * h + alpha * (1 - p) * h
    d = da.reshape(-1, 1)
    dy = d * W2.reshape(1, -1)
    dzh = dy * h * (1 - alpha) * p * (1 - p)
    sums5[:H] = (dy * (1 - p) * h).sum(0)
    sums5[H:2 * H] = dzh.sum(0)
    sums5[2 * H:3 * H] = (dzh * zh).sum(0)
    sums5[3 * H:4 * H] = (d * y).sum(0)
    sums5[4 * H:] = 0
    sums5[4 * H] = d.sum()


def din_attn_bwd(q, K, W1, b1, alpha, eps, training, stats, W2, mask, a_logit, dout, da, sums5,
                 n_total, dq, dK, dW1b1, workspace, dq_accumulate=False):
    B, L, E = K.shape
    H = W1.shape[0]
    x, h = _din_attn_h(q, K, W1, b1)
    rstd, zh, p = _din_attn_gate(h, alpha, eps, stats)
    dy = da.reshape(-1, 1) * W2.reshape(1, -1)
    dzh = dy * h * (1 - alpha) * p * (1 - p)
    if training:
        dzh = dzh - sums5[H:2 * H] / n_total - zh * (sums5[2 * H:3 * H] / n_total)
    dh = dy * (p + alpha * (1 - p)) + dzh * rstd
    dW1b1[:H * 4 * E] = (dh.t() @ x).reshape(-1)
    dW1b1[H * 4 * E:] = dh.sum(0)
    d = (dh @ W1).view(B, L, 4, E)
    dq_new = (d[:, :, 0] + d[:, :, 2] + d[:, :, 3] * K).sum(1)
    dq.copy_(dq + dq_new if dq_accumulate else dq_new)
    wm = a_logit.view(B, L) * _din_attn_mask(mask, B, L)
    dK.copy_(d[:, :, 1] - d[:, :, 2] + d[:, :, 3] * q.unsqueeze(1) + wm.unsqueeze(-1) * dout.unsqueeze(1))


def dot_interact_fwd(emb, F, D, out, tail=0):
    e = emb.reshape(emb.shape[0], -1)[:, :F * D].reshape(-1, F, D)
    P = F * (F - 1) // 2
    ipm = torch.bmm(e, e.transpose(1, 2))
    mask = torch.triu(torch.ones(F, F), 1).bool()
    out[:, :P] = torch.masked_select(ipm, mask).view(-1, P)
    if tail:
        out[:, P:P + D] = e[:, F - 1, :]
        out[:, P + D:P + tail] = 0
    return out


def dot_interact_bwd(emb, g, F, D, demb, tail=0):
    e = emb.reshape(emb.shape[0], -1)[:, :F * D].reshape(-1, F, D)
    P = F * (F - 1) // 2
    G = torch.zeros(e.shape[0], F, F)
    iu = torch.triu_indices(F, F, 1)
    G[:, iu[0], iu[1]] = g[:, :P]
    G = G + G.transpose(1, 2)
    d = torch.bmm(G, e)
    if tail:
        d[:, F - 1, :] += g[:, P:P + D]
    demb.copy_(d.reshape(demb.shape))
    return demb

def cin_workgroups():
    return 4


def cin_wimg_floats(F0, Mi, D, O):
    return 0


def cin_pack_w(layers, D):
    return None


def cin_fwd(X0, Xi, W, bias, Xn, pool, w_img=None):
    had = torch.einsum("bhd,bmd->bhmd", X0, Xi).reshape(X0.shape[0], -1, X0.shape[2])
    out = torch.einsum("oc,bcd->bod", W, had) + bias.view(1, -1, 1)
    Xn.copy_(out)
    if pool is not None:
        pool.copy_(out.sum(-1))


def cin_bwd(X0, Xi, W, dXn, dpool, dX0, accumulate_dx0, dXi, partial, w_img=None):
    B, F0, D = X0.shape
    Mi, O = Xi.shape[1], W.shape[0]
    g = torch.zeros(B, O, D)
    if dXn is not None:
        g = g + dXn
    if dpool is not None:
        g = g + dpool.unsqueeze(-1)
    T = torch.einsum("bod,oc->bcd", g, W).view(B, F0, Mi, D)
    d0 = (T * Xi.unsqueeze(1)).sum(2)
    dX0.copy_(dX0 + d0 if accumulate_dx0 else d0)
    dXi.copy_((T * X0.unsqueeze(2)).sum(1))
    had = torch.einsum("bhd,bmd->bhmd", X0, Xi).reshape(B, -1, D)
    partial.zero_()
    partial[0, :O * F0 * Mi] = torch.einsum("bod,bcd->oc", g, had).reshape(-1)
    partial[0, O * F0 * Mi:] = g.sum((0, 2))


def dedup_sorted_runs(ids, n_runs, vocab, pad, workspace, result=None):
    flat = ids.reshape(-1)
    run_len = flat.numel() // n_runs
    for r in range(n_runs):            # the precondition of the merge: every run ascending
        run = flat[r * run_len:(r + 1) * run_len].long()
        k = torch.where((run >= 0) & (run < vocab) & (run != pad), run, torch.full_like(run, vocab))
        assert bool((k[1:] >= k[:-1]).all()), "run %d is not ascending" % r
    return dedup(flat.view(-1, 1), torch.zeros(1, dtype=torch.int64),
                 torch.tensor([vocab], dtype=torch.int32), torch.tensor([pad], dtype=torch.int32),
                 vocab, workspace, result=result)




# ---- fused sparse front end / back end (csrc/fx_fused.hip) ----------------------------------------
class RowState(object):
    def __init__(self, table, m, v, last_step, D, G=None):
        self.table, self.m, self.v, self.last_step, self.D, self.G = table, m, v, last_step, D, G


def dedup_catchup(ids, col_row_base, col_vocab, col_pad, workspace, states, scal, begin_scal=None,
                  upto_offset=-1, want_uid=False, result=None):
    if begin_scal is not None:
        opt_begin_step(begin_scal)
    # the column fast path keeps padding / bad-id lookups inside their column (key = pad id or 0,
    # position 0xFFFFFFFF = "contributes nothing"): emulated on top of the generic de-dup
    B, C = ids.shape
    keys = ids.long().clamp(min=0)
    bad = (ids < 0) | (ids >= col_vocab.view(1, -1))
    keys = torch.where(bad, torch.zeros_like(keys), keys) + col_row_base.view(1, -1)
    invalid = bad | (ids == col_pad.view(1, -1))
    skey_cols, spos_cols = [], []
    for c in range(C):
        k = keys[:, c]
        # real items first among equal keys, like the in-LDS sort (fill items carry the largest key)
        order = torch.sort(k * 2 + invalid[:, c].long(), stable=True)[1]
        skey_cols.append(k[order])
        pos = torch.arange(B) * C + c
        pos = torch.where(invalid[:, c], torch.full_like(pos, 0xFFFFFFFF), pos)
        spos_cols.append(pos[order])
    skey, spos = torch.cat(skey_cols), torch.cat(spos_cols)
    uniq, counts = torch.unique_consecutive(skey, return_counts=True)
    dd = DedupResult()
    n = B * C
    dd.sorted_key = skey.int()
    dd.sorted_pos = spos          # int64 holding 0xFFFFFFFF for "no position"
    dd.uniq_row = torch.zeros(n, dtype=torch.int64)
    dd.uniq_row[:len(uniq)] = uniq
    dd.seg_start = torch.zeros(n + 1, dtype=torch.int32)
    dd.seg_start[1:len(uniq) + 1] = torch.cumsum(counts, 0).int()
    dd.n_unique = torch.tensor([len(uniq)], dtype=torch.int32)
    dd.n_max, dd.C = n, C
    dd.sorted_uid = torch.repeat_interleave(torch.arange(len(uniq)), counts).int()
    for st in states:
        adam_catchup(st.table, st.m, st.v, st.last_step, st.D, dd, st.table.shape[0], upto_offset,
                     scal)
    return dd


def emb_fm_fwd(table, D, ids, col_row_base, col_vocab, col_out_off, dense, num_w, num_out_off, out,
               scal, table1=None, num_w1=None, bias1=None, lr_out=None, fm_out=None, fm_lr_out=None,
               S=None, zero_ranges=()):
    emb_gather_fwd(table, D, ids, col_row_base, col_vocab, col_out_off, dense, num_w, num_out_off,
                   out, scal)
    for off, n in zero_ranges:
        out[:, off:off + n] = 0
    lr = None
    if lr_out is not None or fm_lr_out is not None:
        lr = torch.empty(out.shape[0], 1)
        lr_fwd(table1, ids, col_row_base, col_vocab, dense, num_w1, bias1, lr, scal)
        if lr_out is not None:
            lr_out.copy_(lr)
    if fm_out is not None or fm_lr_out is not None or S is not None:
        e = out.view(out.shape[0], -1, D)
        s = e.sum(1)
        fm = 0.5 * ((s ** 2) - (e ** 2).sum(1)).sum(-1, keepdim=True)
        if S is not None:
            S.copy_(s)
        if fm_out is not None:
            fm_out.copy_(fm)
        if fm_lr_out is not None:
            fm_lr_out.copy_(fm + lr)
    return out


def emb_fm_bwd_partials(n_lookups, D):
    return 4


def emb_fm_bwd_workspace_floats(n_lookups, D, Fd):
    return 16


def emb_fm_bwd(drec, rec, S, g_fm, g_lr, col_out_off, C, D, dd, G, sq_partials, G1, sq1_partials,
               dense, num_out_off, B, dnum_w, dnum_w1, dbias1, workspace=None):
    if rec is not None:
        full = torch.zeros(B, rec.shape[1])
    else:
        full = torch.zeros(B, drec.shape[1])
    if drec is not None:
        full = full + drec
    if g_fm is not None:
        e = rec.view(B, -1, D)
        full = full + (g_fm.view(B, 1, 1) * (S.view(B, 1, D) - e)).reshape(B, -1)
    if C > 0 and dd is not None and dd.n_max > 0:
        nu = int(dd.n_unique)
        G.zero_()
        sq_partials.zero_()
        if G1 is not None:
            G1.zero_()
            sq1_partials.zero_()
        for u in range(nu):
            for i in range(int(dd.seg_start[u]), int(dd.seg_start[u + 1])):
                p = int(dd.sorted_pos[i])
                if p == 0xFFFFFFFF:
                    continue
                b, c = divmod(p, C)
                o = int(col_out_off[c])
                G[u] += full[b, o:o + D]
                if G1 is not None:
                    G1[u] += g_lr[b, 0]
        sq_partials[0] = float((G[:nu].double() ** 2).sum())
        if G1 is not None:
            sq1_partials[0] = float((G1[:nu].double() ** 2).sum())
    if dense is not None:
        for j in range(dense.shape[1]):
            o = int(num_out_off[j])
            dnum_w[j] = (dense[:, j:j + 1] * full[:, o:o + D]).sum(0)
            if dnum_w1 is not None:
                dnum_w1[j] = (dense[:, j:j + 1] * g_lr).sum()
    if dbias1 is not None and g_lr is not None:
        dbias1[0] = g_lr.sum()


def adam_catchup_all(state, total_rows, upto_offset, scal):
    adam_catchup(state.table, state.m, state.v, state.last_step, state.D, None, total_rows,
                 upto_offset, scal)


def adam_catchup_rows(states, dd, upto_offset, scal):
    for st in states:
        adam_catchup(st.table, st.m, st.v, st.last_step, st.D, dd, st.table.shape[0], upto_offset,
                     scal)


def sparse_update_multi(kind, states, dd, scal):
    for st in states:
        if kind == "adam":
            sparse_adam(st.table, st.m, st.v, st.last_step, st.D, dd, st.G, scal)
        else:
            sparse_sgd(st.table, st.D, dd, st.G, scal, last_step=st.last_step)


def pack_columns_multi(items):
    for t, out, col0 in items:
        t2 = t.reshape(t.shape[0], -1)
        out[:, col0:col0 + t2.shape[1]] = t2.to(out.dtype)


class KernelTimer(object):
    recording = False


NAMES = ["new_scalars", "pack_columns", "emb_gather_fwd", "dedup_workspace_bytes", "dedup",
         "emb_grad_reduce_partials", "emb_grad_reduce_scratch_ints", "emb_grad_reduce",
         "emb_numeric_grad", "opt_begin_step", "clip_coef", "sparse_adam", "adam_catchup",
         "sparse_sgd", "mt_sqnorm", "mt_adam", "mt_sgd", "fm_fwd", "fm_bwd", "lr_fwd", "gemm",
         "colsum", "mask_mul", "cross_bwd_prep", "sigmoid_bce", "shard_plan", "scatter_rows",
         "sum_parts", "din_concat_fwd", "din_concat_bwd", "din_pool_fwd", "din_pool_bwd",
         "dice_workspace_floats", "dice_fwd", "dice_bwd", "dot_interact_fwd", "dot_interact_bwd",
         "cin_workgroups", "cin_fwd", "cin_bwd", "cin_wimg_floats", "cin_pack_w", "reg_stats", "reg_cross", "reg_dense_update",
         "shard_plan_workspace_ints", "emb_seq_pool_fwd", "dedup_sorted_runs", "RowState",
         "dedup_catchup", "emb_fm_fwd", "emb_fm_bwd", "sparse_update_multi", "pack_columns_multi",
         "emb_fm_bwd_partials", "emb_fm_bwd_workspace_floats", "adam_catchup_all", "adam_catchup_rows",
         "dice_local_sums", "dice_fwd_from_sums", "dice_bwd_local_sums", "dice_bwd_from_sums",
         "din_attn_workspace_floats", "din_attn_stats", "dice_stats_from_sums", "din_attn_fwd",
         "din_attn_bwd_sums", "din_attn_bwd", "gemm_dw_dx", "split_rows", "gemm_problem", "gemm_batch",
         "gemm_workspace_floats", "fill_grad_block", "owner_grad_reduce", "owner_grad_reduce_partials",
         "owner_fetch_rows", "head_train", "head_train_ok", "head_train_workspace_floats"]


def install_plain():
    """Same as install() without pytest's monkeypatch (for spawned worker processes)."""
    import fuxictr_amd.ops as real
    import fuxictr_amd.optim as optim
    import fuxictr_amd.rank_model as rm
    me = globals()
    for name in NAMES:
        setattr(real, name, me[name])
    rm.get_device = lambda gpu=-1: torch.device("cpu")
    torch.cuda.set_device = lambda d: None
    optim._NativeOptimizer._require_cuda = False


def install(monkeypatch):
    """Route fuxictr_amd.ops.* to this module and let BaseModel live on the CPU."""
    import fuxictr_amd.ops as real
    import fuxictr_amd.rank_model as rm
    me = globals()
    for name in NAMES:
        monkeypatch.setattr(real, name, me[name])
    monkeypatch.setattr(rm, "get_device", lambda gpu=-1: torch.device("cpu"))
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
