"""Training-mode forward (activations kept) and backward of the VGGish CNN as HIP kernel calls
(vggish.py:21-31 forward; its autograd backward in the reference, train.py:137 after
train.py:96-97 made the CNN trainable). f32 only (exact MFMA), NHWC.

Forward keeps, per conv layer, its input and its pre-pool post-ReLU output; the pooled layers
run as conv (pool = 0) + mla_maxpool2x2 so that the pool/ReLU backward can route gradients to
the first maximum of each window. Backward per layer: mla_relu_pool_bwd -> dZ; mla_conv_wgrad
(dW), column sums (db); mla_conv3x3 with flipped/transposed weights (dgrad). The three Linear
layers use the MFMA GEMM on transposed copies, as the MLA head does."""

import torch

from . import mla_train, ops

#            cin, cout,  H,  W, pooled     (features[0] is handled by the fused conv1 kernels)
GEOM = {2: (64, 128, 48, 32, True), 3: (128, 256, 24, 16, False), 4: (256, 256, 24, 16, True),
        5: (256, 512, 12, 8, False), 6: (512, 512, 12, 8, True)}


def _parts(cnn_model):
    """(features container, list of Linear holders or [], key prefixes) for VGGish or the
    just_bottlenecks Sequential(features, CnnFlatten)."""
    if hasattr(cnn_model, "features"):
        return cnn_model.features, cnn_model.embeddings._fcs, "features.", "embeddings."
    return cnn_model[0], [], "0.", None


def forward(cnn_model, x):
    """x: (N, 1, 96, 64) or (N, 96, 64) float32 -> ((N, 128) embeddings or (N, 12288) bottlenecks, tape)."""
    feats, fcs, _, _ = _parts(cnn_model)
    convs = feats._convs
    x = x.detach().reshape(-1, 96, 64).float().contiguous()
    tape = {"x": x, "layers": {}}
    cur = ops.conv1(x, convs[0].weight.detach().contiguous(), convs[0].bias.detach(), torch.float32)
    for layer in range(2, 7):
        cin, cout, H, W_, pooled = GEOM[layer]
        c = convs[layer - 1]
        wp = ops.repack_conv_weight(c.weight.detach().contiguous(), torch.float32)
        a = ops.conv3x3(cur, wp, c.bias.detach(), cout, pool=False, act=True)
        tape["layers"][layer] = (cur, a)
        cur = ops.maxpool2x2(a) if pooled else a
    h = cur.reshape(cur.shape[0], -1)
    tape["fc"] = []
    for f in fcs:
        out = ops.linear(h, f.weight.detach(), f.bias.detach(), relu=True)
        tape["fc"].append((h, out))
        h = out
    return h, tape


def backward(cnn_model, tape, d_out, grads, prefix):
    """d_out: gradient w.r.t. forward()'s result. Fills grads[prefix + <state_dict key>] for every CNN
    parameter (weights in state_dict layout)."""
    feats, fcs, kf, ke = _parts(cnn_model)
    convs = feats._convs
    conv_idx = [0, 3, 6, 8, 11, 13]
    fc_idx = [0, 2, 4]
    d = d_out.contiguous()
    for i in range(len(fcs) - 1, -1, -1):
        h_in, h_out = tape["fc"][i]
        dz = ops.relu_pool_bwd(h_out, d, pool=False)
        key = prefix + ke + "%d." % fc_idx[i]
        d = mla_train._linear_backward(h_in, fcs[i].weight.detach(), dz, grads[key + "weight"], grads[key + "bias"], True)
    n = tape["x"].shape[0]
    d = d.reshape(n, 6, 4, 512)
    for layer in range(6, 1, -1):
        cin, cout, H, W_, pooled = GEOM[layer]
        a_in, a = tape["layers"][layer]
        key = prefix + kf + "%d." % conv_idx[layer - 1]
        dz = ops.relu_pool_bwd(a, d.contiguous(), pool=pooled, db=grads[key + "bias"])     # bias gradient summed on the way
        ops.conv_wgrad(dz, a_in, grads[key + "weight"])
        wd = ops.repack_dgrad(convs[layer - 1].weight.detach().contiguous())
        d = ops.conv3x3(dz, wd, None, cin, pool=False, act=False)
    key = prefix + kf + "0."
    ops.conv1_bwd(tape["x"], convs[0].weight.detach().contiguous(), convs[0].bias.detach(), d.contiguous(),
                  grads[key + "weight"], grads[key + "bias"])
