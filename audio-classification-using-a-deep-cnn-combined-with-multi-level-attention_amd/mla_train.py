"""Forward (and, for training, backward) of the multi-level-attention head as sequences of
HIP kernel calls (model.py:217-222, :236-242, :258-269). Shared by the drop-in modules of
``model.py`` and by the fused training step (``train.py``)."""

import torch

from . import ops
from .params import DR, H, K, T


def _bn_stats(bn, z, mode, period, train):
    """(mean, var) to normalise with: batch statistics in train mode (running buffers updated
    in place, torch semantics), running statistics in eval mode."""
    if train:
        mean, var = ops.bn_stats(z, mode, period, bn.running_mean, bn.running_var, bn.momentum)
        bn.num_batches_tracked += 1
        return mean, var
    return bn.running_mean, bn.running_var


def embedded_mapping_forward(em, x, tape):
    """model.py:217-222 on x (B, T, F) float32 -> (B, T, H)."""
    B = x.shape[0]
    rows = x.detach().reshape(B * T, -1).float().contiguous()
    train = em.training
    mean, var = _bn_stats(em.norm0, rows, 0, T, train)
    h = ops.bn_apply(rows, 0, T, mean, var, em.norm0.weight.detach(), em.norm0.bias.detach())
    if tape is not None:
        tape.append(("norm0", em, rows, mean, var))
    for j in range(em.n_fc):
        z = ops.linear(h, em.fc[j].weight.detach(), em.fc[j].bias.detach())
        mean, var = _bn_stats(em.norms[j], z, 0, T, train)
        keep = em.dropouts[j].keep_mask(z.numel(), z.device) if train else None
        h_in = h
        h = ops.bn_apply(z, 0, T, mean, var, em.norms[j].weight.detach(), em.norms[j].bias.detach(), act=1,
                         keep_mask=keep, drop_scale=1.0 / (1.0 - DR))
        if tape is not None:
            tape.append(("fc", em, j, h_in, z, mean, var, keep, h))
    return h.reshape(B, T, H)


def attention_forward(am, h, y, tape):
    """model.py:236-242 on h (B, T, H): writes y (B, K) (a column block of the concatenation)."""
    B = h.shape[0]
    rows = h.reshape(B * T, H)
    z = ops.linear(rows, am.fcv.weight.detach(), am.fcv.bias.detach())
    train = am.training
    if train:
        # both norms see the same z, hence the same batch statistics; each updates its own buffers
        mean, var = ops.bn_stats(z, 0, T, am.normv.running_mean, am.normv.running_var, am.normv.momentum)
        ops.bn_stats(z, 0, T, am.normf.running_mean, am.normf.running_var, am.normf.momentum)
        am.normv.num_batches_tracked += 1
        am.normf.num_batches_tracked += 1
        nv = (mean, var, am.normv.weight.detach(), am.normv.bias.detach())
        nf = (mean, var, am.normf.weight.detach(), am.normf.bias.detach())
    else:
        nv = (am.normv.running_mean, am.normv.running_var, am.normv.weight.detach(), am.normv.bias.detach())
        nf = (am.normf.running_mean, am.normf.running_var, am.normf.weight.detach(), am.normf.bias.detach())
    att, cla = ops.attention_pool(z, B, T, K, nv, nf, y, save=tape is not None)
    if tape is not None:
        tape.append(("att", am, rows, z, nv, nf, att, cla))


def mla_forward(mla, x, tape=None):
    """model.py:258-269: x (B, T, M) -> (B, K) sigmoid scores."""
    B = x.shape[0]
    L = len(mla.model)
    conc = torch.empty((B, L * K), dtype=torch.float32, device=x.device)
    cur = x
    for lvl in range(L):
        cur = embedded_mapping_forward(mla.embedded_mappings[lvl], cur, tape)
        attention_forward(mla.attention_modules[lvl], cur, conc[:, lvl * K:(lvl + 1) * K], tape)
    z = ops.linear_small(conc, mla.fc.weight.detach(), mla.fc.bias.detach())
    mean, var = _bn_stats(mla.norm, z, 1, 0, mla.training)
    out = ops.bn_apply(z, 1, 0, mean, var, mla.norm.weight.detach(), mla.norm.bias.detach(), act=2)
    if tape is not None:
        tape.append(("head", mla, conc, z, mean, var, out))
    return out


def mla_apply(mla, x):
    """Module-level entry: inference / train-mode forward without autograd history."""
    return mla_forward(mla, x, None)
