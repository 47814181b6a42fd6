"""Forward and backward of the multi-level-attention head as sequences of HIP kernel calls
(model.py:217-222, :236-242, :258-269). Shared by the drop-in modules of ``model.py`` (forward
only) and by the fused training step of ``train.py`` (forward with a tape, then backward)."""

import torch

from . import ops
from .params import DR, H, K, T


class Ctx:
    """Per-step context: the tape of saved activations and the data-parallel group (SyncBN)."""

    def __init__(self, tape=False, dist=None, counter=None, bases=None):
        """counter / bases: set while the training step is being captured into a HIP graph -- a device int64 holding the number of
        completed steps and {id(Dropout module): its call count minus that number}; masks are then drawn by call number on the device."""
        self.tape = [] if tape else None
        self.dist = dist or ops._local()
        self.counter, self.bases = counter, bases
        self.train = True


def _bn_stats(bn, z, mode, period, train, ctx):
    """(mean, var) to normalise with: batch statistics in train mode (running buffers updated in
    place, torch semantics; sums all-reduced over the data-parallel group), else running ones."""
    if train:
        return ops.bn_stats_sync(z, mode, period, ctx.dist, bn.running_mean, bn.running_var, bn.momentum, tracked=bn.num_batches_tracked)
    return bn.running_mean, bn.running_var


def embedded_mapping_forward(em, x, ctx=None):
    """model.py:217-222 on x (B, T, F) float32 -> (B, T, H)."""
    ctx = ctx or Ctx()
    B = x.shape[0]
    rows = x.detach().reshape(B * T, -1).float().contiguous()
    train = em.training
    mean, var = _bn_stats(em.norm0, rows, 0, T, train, ctx)
    h = ops.bn_apply(rows, 0, T, mean, var, em.norm0.weight.detach(), em.norm0.bias.detach())
    if ctx.tape is not None:
        ctx.tape.append(("norm0", em, rows, mean, var))
    for j in range(em.n_fc):
        z = ops.linear(h, em.fc[j].weight.detach(), em.fc[j].bias.detach())
        mean, var = _bn_stats(em.norms[j], z, 0, T, train, ctx)
        if train and ctx.counter is not None:
            d = em.dropouts[j]
            keep = d.keep_mask_dev(z.numel(), z.device, ctx.dist.rank * z.numel(), ctx.counter, ctx.bases[id(d)])
        else:
            keep = em.dropouts[j].keep_mask(z.numel(), z.device, offset=ctx.dist.rank * z.numel()) if train else None
        h_in = h
        h = ops.bn_apply(z, 0, T, mean, var, em.norms[j].weight.detach(), em.norms[j].bias.detach(), act=1,
                         keep_mask=keep, drop_scale=1.0 / (1.0 - DR))
        if ctx.tape is not None:
            ctx.tape.append(("fc", em, j, h_in, z, mean, var, h))
    return h.reshape(B, T, H)


def attention_forward(am, h, y, ctx=None, level=0):
    """model.py:236-242 on h (B, T, H): writes y (B, K) (a column block of the concatenation)."""
    ctx = ctx or Ctx()
    B = h.shape[0]
    rows = h.reshape(B * T, H)
    z = ops.linear(rows, am.fcv.weight.detach(), am.fcv.bias.detach())
    if am.training:
        # both norms see the same z, hence the same batch statistics; each keeps its own buffers
        mean, var = ops.bn_stats_sync(z, 0, T, ctx.dist, am.normv.running_mean, am.normv.running_var, am.normv.momentum,
                                      also=((am.normf.running_mean, am.normf.running_var, am.normf.momentum, am.normf.num_batches_tracked),),
                                      tracked=am.normv.num_batches_tracked)
        nv = (mean, var, am.normv.weight.detach(), am.normv.bias.detach())
        nf = (mean, var, am.normf.weight.detach(), am.normf.bias.detach())
    else:
        nv = (am.normv.running_mean, am.normv.running_var, am.normv.weight.detach(), am.normv.bias.detach())
        nf = (am.normf.running_mean, am.normf.running_var, am.normf.weight.detach(), am.normf.bias.detach())
    att, cla = ops.attention_pool(z, B, T, K, nv, nf, y, save=ctx.tape is not None)
    if ctx.tape is not None:
        ctx.tape.append(("att", am, level, rows, z, nv, nf, att, cla))


def mla_forward(mla, x, ctx=None):
    """model.py:258-269: x (B, T, M) -> (B, K) sigmoid scores."""
    ctx = ctx or Ctx()
    ctx.train = mla.training                 # batch statistics + dropout (train) or running statistics (eval): the backward must match
    B = x.shape[0]
    L = len(mla.model)
    conc = torch.empty((B, L * K), dtype=torch.float32, device=x.device)
    cur = x
    for lvl in range(L):
        cur = embedded_mapping_forward(mla.embedded_mappings[lvl], cur, ctx)
        attention_forward(mla.attention_modules[lvl], cur, conc[:, lvl * K:(lvl + 1) * K], ctx, lvl)
    z = ops.linear_small(conc, mla.fc.weight.detach(), mla.fc.bias.detach())
    mean, var = _bn_stats(mla.norm, z, 1, 0, mla.training, ctx)
    out = ops.bn_apply(z, 1, 0, mean, var, mla.norm.weight.detach(), mla.norm.bias.detach(), act=2)
    if ctx.tape is not None:
        ctx.tape.append(("head", mla, conc, z, mean, var, out))
    return out


def mla_apply(mla, x):
    """Module-level entry: inference / train-mode forward (no tape)."""
    return mla_forward(mla, x, None)


def _linear_backward(x_in, weight, dz, g_w, g_b, want_dx):
    """out = x_in W^T + b: dW = dz^T x_in, db = column sums, dx = dz W -- MFMA GEMMs on transposed,
    K-contiguous copies (reduction over rows for dW, over out-features for dx). g_w / g_b None: that gradient is not wanted."""
    if g_w is not None:
        dzT = ops.transpose_padded(dz)                         # (N, M~)
        xT = ops.transpose_padded(x_in)                        # (Kin, M~)
        ops.linear(dzT, xT, None, out=g_w, split_k=True)
    if g_b is not None:
        ops.col_sum(dz, g_b)
    if not want_dx:
        return None
    wT = ops.transpose_padded(weight)                      # (Kin, N~)
    if dz.shape[1] % 4:
        return ops.linear_small(dz, wT[:, :dz.shape[1]].contiguous(), None)
    return ops.linear(dz, wT, None)


def mla_backward(mla, ctx, dout, grads, need_input_grad=False):
    """Backward of mla_forward from d(loss)/d(out) = dout (B, K). `grads` maps parameter names
    relative to `mla` (e.g. 'embedded_mappings.0.fc.1.weight') to preallocated gradient tensors,
    all of which are overwritten. Returns d(loss)/dx (B*T, M) if need_input_grad."""
    tape, dist = ctx.tape, ctx.dist
    bs = ctx.train                           # eval-mode forward: fixed (running) statistics, no dropout scale
    drop = 1.0 / (1.0 - DR) if bs else 1.0
    kind, _, conc, z, mean, var, out = tape[-1]
    assert kind == "head"
    dz = ops.bn_backward(z, dout, out, 2, 1.0, 1, 0, mean, var, mla.norm.weight.detach(), dist, grads["norm.weight"], grads["norm.bias"],
                         batch_stats=bs)
    dconc = ops.linear_small_bwd(conc, mla.fc.weight.detach(), dz, grads["fc.weight"], grads["fc.bias"])

    # split the tape per level
    levels, cur = [], []
    for e in tape[:-1]:
        cur.append(e)
        if e[0] == "att":
            levels.append(cur)
            cur = []
    dh_next = None                      # gradient reaching this level's output from the level above
    B = dconc.shape[0]
    for lvl in range(len(levels) - 1, -1, -1):
        entries = levels[lvl]
        _, am, _, h_rows, z_att, nv, nf, att, cla = entries[-1]
        pa = "attention_modules.%d." % lvl
        du_v, du_f = ops.attention_pool_bwd(dconc[:, lvl * K:(lvl + 1) * K], att, cla, B, T, K)
        dz_att = ops.bn_backward(z_att, du_v, None, 0, 1.0, 0, T, nv[0], nv[1], nv[2], dist, grads[pa + "normv.weight"], grads[pa + "normv.bias"],
                                 batch_stats=bs)
        ops.bn_backward(z_att, du_f, None, 0, 1.0, 0, T, nf[0], nf[1], nf[2], dist, grads[pa + "normf.weight"], grads[pa + "normf.bias"],
                        dx=dz_att, accumulate=True, batch_stats=bs)
        dh = _linear_backward(h_rows, am.fcv.weight.detach(), dz_att, grads[pa + "fcv.weight"], grads[pa + "fcv.bias"], True)
        if dh_next is not None:
            ops.axpy(1.0, dh_next, dh)
        em = entries[0][1]
        pe = "embedded_mappings.%d." % lvl
        for e in reversed(entries[1:-1]):
            _, _, j, h_in, z_j, mean_j, var_j, h_out = e
            dzj = ops.bn_backward(z_j, dh, h_out, 1, drop, 0, T, mean_j, var_j, em.norms[j].weight.detach(), dist,
                                  grads[pe + "norms.%d.weight" % j], grads[pe + "norms.%d.bias" % j], batch_stats=bs)
            dh = _linear_backward(h_in, em.fc[j].weight.detach(), dzj, grads[pe + "fc.%d.weight" % j], grads[pe + "fc.%d.bias" % j], True)
        _, _, rows_in, mean0, var0 = entries[0]
        want = lvl > 0 or need_input_grad
        dh_next = ops.bn_backward(rows_in, dh, None, 0, 1.0, 0, T, mean0, var0, em.norm0.weight.detach(), dist,
                                  grads[pe + "norm0.weight"], grads[pe + "norm0.bias"], want_dx=want, batch_stats=bs)
    return dh_next
