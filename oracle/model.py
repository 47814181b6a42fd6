"""ORACLE (test infrastructure, not product code): PyTorch-CPU float32 restatement of
the reference's VGGish + multi-level-attention model and of its training step.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module.

Pinned: ``tests/test_oracle_golden.py`` checks it against
``tests/golden/model_*.npz`` produced by ``tests/golden/make_golden.py`` from the
reference's own ``model.Ensemble`` / ``VGGish`` classes (imported from
``/root/reference`` in the build container) on portable-seeded weights.

Everything is written functionally over a plain ``{key: tensor}`` state dict whose
keys are the reference's ``state_dict`` keys, so nothing here shares structure with
the reference's nn.Module classes. Citations are relative to ``/root/reference``.
"""

import numpy as np
import torch
import torch.nn.functional as F

T, H, K, DR = 10, 600, 10, 0.4          # params.py:26-32
BN_EPS, BN_MOMENTUM = 1e-5, 0.1         # torch.nn.BatchNorm1d defaults (model.py:205, 213, 232-233, 256)
CONV_IDX = (0, 3, 6, 8, 11, 13)         # vggish.py:108-118 Sequential positions of the convs
POOL_AFTER = (0, 3, 8, 13)              # convs followed by MaxPool2d(2, 2)
FC_IDX = (0, 2, 4)                      # vggish.py:13-19


def to_torch(sd):
    return {k: torch.as_tensor(np.asarray(v)) for k, v in sd.items()}


# ------------------------------------------------------------------ VGGish ---

def vgg_features(sd, x, prefix="features.", taps=None):
    """vggish.py:108-118 + :22 -- (N,1,96,64) -> (N,512,6,4); conv3x3 pad1 + ReLU, 4 max-pools."""
    for idx in CONV_IDX:
        x = F.relu(F.conv2d(x, sd["%s%d.weight" % (prefix, idx)], sd["%s%d.bias" % (prefix, idx)], padding=1))
        if idx in POOL_AFTER:
            x = F.max_pool2d(x, 2, 2)
        if taps is not None:
            taps.append(x)
    return x


def nhwc_flatten(x):
    """vggish.py:26-29 / model.py:190-193 -- NCHW -> (N, h*w*c) with index (h*W + w)*C + c."""
    return x.permute(0, 2, 3, 1).contiguous().reshape(x.shape[0], -1)


def vgg_embeddings(sd, x, prefix="embeddings."):
    """vggish.py:13-19 + :31 -- three Linear+ReLU, the last ReLU included."""
    for idx in FC_IDX:
        x = F.relu(F.linear(x, sd["%s%d.weight" % (prefix, idx)], sd["%s%d.bias" % (prefix, idx)]))
    return x


def vggish_forward(sd, x, prefix=""):
    """VGGish.forward with preprocess=False, postprocess=False (vggish.py:166-172, 21-31)."""
    return vgg_embeddings(sd, nhwc_flatten(vgg_features(sd, x, prefix + "features.")), prefix + "embeddings.")


def postprocess(pca_eigen_vectors, pca_means, emb):
    """vggish.py:62-102 -- PCA, clamp to [-2, 2], quantise to 0..255 (still float)."""
    y = torch.mm(pca_eigen_vectors, emb.t() - pca_means).t()
    y = torch.clamp(y, -2.0, 2.0)
    return torch.squeeze(torch.round((y + 2.0) * (255.0 / 4.0)))


# --------------------------------------------------------------------- MLA ---

def bn_t(sd, key, x, train, stats_out=None):
    """BatchNorm1d(C) on (B, C, L) or (B, C): channel axis 1, statistics over the rest.

    Train mode uses the biased batch variance for normalisation and updates the
    running statistics with the unbiased one (torch semantics, momentum 0.1).
    """
    w, b = sd[key + ".weight"], sd[key + ".bias"]
    red = [d for d in range(x.dim()) if d != 1]
    shape = [1, -1] + [1] * (x.dim() - 2)
    if train:
        mean = x.mean(dim=red)
        var = x.var(dim=red, unbiased=False)
        if stats_out is not None:
            n = x.numel() // x.shape[1]
            stats_out[key] = (mean.detach().clone(), (var.detach() * n / max(n - 1, 1)).clone())
    else:
        mean, var = sd[key + ".running_mean"], sd[key + ".running_var"]
    return (x - mean.view(shape)) / torch.sqrt(var.view(shape) + BN_EPS) * w.view(shape) + b.view(shape)


def embedded_mapping(sd, p, x, n_fc, train, masks, stats_out):
    """model.py:217-222 -- BN(T) then n_fc x [Linear -> BN(T) -> ReLU -> Dropout(0.4)]."""
    x = bn_t(sd, p + "norm0", x, train, stats_out)
    for j in range(n_fc):
        x = F.linear(x, sd[p + "fc.%d.weight" % j], sd[p + "fc.%d.bias" % j])
        x = F.relu(bn_t(sd, p + "norms.%d" % j, x, train, stats_out))
        if train:
            m = masks[p + "dropouts.%d" % j].to(x.dtype).reshape(x.shape)
            x = x * m / (1.0 - DR)
    return x


def attention_module(sd, p, h, train, stats_out):
    """model.py:236-242 -- fcv feeds BOTH branches; fcf exists but is never used."""
    z = F.linear(h, sd[p + "fcv.weight"], sd[p + "fcv.bias"])
    att = torch.softmax(bn_t(sd, p + "normv", z, train, stats_out), dim=2)
    cla = torch.sigmoid(bn_t(sd, p + "normf", z, train, stats_out))
    norm_att = att / att.sum(dim=1, keepdim=True)
    return (cla * norm_att).sum(dim=1)


def mla_forward(sd, x, model_conf=(2, 1), train=False, masks=None, stats_out=None, prefix="mla."):
    """model.py:258-269 -- (B, T, M) -> (B, K) in (0, 1)."""
    embs = []
    cur = x
    for lvl, n_fc in enumerate(model_conf):
        cur = embedded_mapping(sd, "%sembedded_mappings.%d." % (prefix, lvl), cur, n_fc, train, masks, stats_out)
        embs.append(cur)
    ys = [attention_module(sd, "%sattention_modules.%d." % (prefix, lvl), embs[lvl], train, stats_out)
          for lvl in range(len(model_conf))]
    conc = torch.cat(ys, dim=1)
    out = F.linear(conc, sd[prefix + "fc.weight"], sd[prefix + "fc.bias"])
    return torch.sigmoid(bn_t(sd, prefix + "norm", out, train, stats_out))


def dropout_mask_keys(model_conf=(2, 1), prefix="mla."):
    return ["%sembedded_mappings.%d.dropouts.%d" % (prefix, lvl, j)
            for lvl, n_fc in enumerate(model_conf) for j in range(n_fc)]


# ---------------------------------------------------------------- Ensemble ---

def ensemble_forward(sd, x, model_conf=(2, 1), just_bottlenecks=False, train=False, masks=None,
                     stats_out=None):
    """model.py:58-62 -- Input reshape (:98-99) -> CNN (:172-175) -> reshape(-1, T, emb) -> MLA."""
    x = x.reshape(-1, 1, 96, 64)
    if just_bottlenecks:
        feats = nhwc_flatten(vgg_features(sd, x, "cnn.cnn_model.0."))
    else:
        feats = vggish_forward(sd, x, "cnn.cnn_model.")
    return mla_forward(sd, feats.reshape(-1, T, feats.shape[1]), model_conf, train, masks, stats_out)


def trainable_keys(sd, finetune=False):
    """train.py:283-303 with model.py:159-160 -- float parameters with requires_grad.

    CNN parameters are frozen unless finetune (train.py:96-97). Buffers
    (running_mean/var, num_batches_tracked) are never parameters.
    """
    keys = []
    for k in sd:
        leaf = k.rsplit(".", 1)[-1]
        if leaf in ("running_mean", "running_var", "num_batches_tracked"):
            continue
        if k.startswith("cnn.") and not finetune:
            continue
        keys.append(k)
    return keys


def apply_running_stats(sd, stats, momentum=BN_MOMENTUM):
    for key, (mean, var_unbiased) in stats.items():
        sd[key + ".running_mean"] = (1 - momentum) * sd[key + ".running_mean"] + momentum * mean
        sd[key + ".running_var"] = (1 - momentum) * sd[key + ".running_var"] + momentum * var_unbiased
        sd[key + ".num_batches_tracked"] = sd[key + ".num_batches_tracked"] + 1


class TrainState:
    """The reference's inner training step (train.py:119-142) on a functional state dict:
    zero_grad -> forward (train mode) -> CrossEntropyLoss on the sigmoid outputs
    (train.py:372, :131) -> backward -> Adam (train.py:369; lr 1e-3, betas 0.9/0.999,
    eps 1e-8). ``fcf.*`` never receive a gradient (model.py:237-238) and are skipped by
    Adam exactly as torch skips ``grad is None`` parameters."""

    def __init__(self, sd, model_conf=(2, 1), just_bottlenecks=False, finetune=False, lr=1e-3):
        self.sd = {k: v.clone() for k, v in to_torch(sd).items()}
        self.model_conf, self.jb, self.finetune = tuple(model_conf), just_bottlenecks, finetune
        self.keys = trainable_keys(self.sd, finetune)
        for k in self.keys:
            self.sd[k].requires_grad_(True)
        self.opt = torch.optim.Adam([self.sd[k] for k in self.keys], lr=lr)

    def step(self, x, labels, masks):
        self.opt.zero_grad()
        stats = {}
        out = ensemble_forward(self.sd, x, self.model_conf, self.jb, train=True, masks=masks, stats_out=stats)
        loss = F.cross_entropy(out, labels)
        loss.backward()
        grads = {k: (None if self.sd[k].grad is None else self.sd[k].grad.detach().clone()) for k in self.keys}
        self.opt.step()
        with torch.no_grad():
            apply_running_stats(self.sd, stats)
        return float(loss.item()), out.detach(), grads
