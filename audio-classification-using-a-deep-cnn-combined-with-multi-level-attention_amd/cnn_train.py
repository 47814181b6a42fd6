"""Training-mode forward (activations kept) and backward of the VGGish CNN as HIP kernel calls
(vggish.py:21-31 forward; its autograd backward in the reference, train.py:137 after
train.py:96-97 made the CNN trainable). NHWC. Two arithmetic modes, chosen by the CNN's ``precision``:

* "f32": exact f32 MFMA everywhere (parity mode: gradients within 1e-4 of the reference's autograd);
* "bf16": bf16 activations, gradients and weight copies with f32 accumulation; weight and bias gradients are
  produced in f32 and applied to the f32 master parameters (mixed precision; bf16 has f32's exponent range, so no
  loss scaling). The bf16 weight copies are re-derived from the masters after every Adam step.

Forward keeps, per conv layer, its input and what the pool / ReLU backward needs of its output: the un-pooled layers their
post-ReLU output, the pooled layers one byte per pooled element in bf16 mode (mla_conv3x3_train_codes: the window position of the
first maximum, or "ReLU off") and the whole pre-pool activation in f32 mode (mla_conv3x3_train), so that the backward can route
gradients to the first maximum of each window. Backward per layer: mla_relu_pool_bwd -> dZ (+ db on the way); mla_conv_wgrad
(dW); mla_conv3x3 with flipped/transposed weights (dgrad). The three Linear
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


def features_forward(feats, x, dtype):
    """Training forward of the conv stack (vggish.py:108-118): x (N, 1, 96, 64) or (N, 96, 64) float32 -> ((N, 6, 4, 512) NHWC in
    `dtype`, tape). The tape keeps, per conv layer, its input and what its pool / ReLU backward needs (module docstring)."""
    convs = feats._convs
    x = x.detach().reshape(-1, 96, 64).float().contiguous()
    tape = {"x": x, "layers": {}, "dtype": dtype}
    cur = ops.conv1(x, convs[0].weight.detach().contiguous(), convs[0].bias.detach(), dtype)
    packed = feats._cache.get([c.weight for c in convs[1:]], dtype,
                              lambda: [ops.repack_conv_weight(c.weight.detach().contiguous(), dtype) for c in convs[1:]])
    for layer in range(2, 7):
        cin, cout, H, W_, pooled = GEOM[layer]
        c = convs[layer - 1]
        if pooled and dtype == torch.bfloat16:      # one kernel writes the pooled activation and one byte per pooled element (window
            a, nxt = ops.conv3x3_train_codes(cur, packed[layer - 2], c.bias.detach(), cout)      # position of the maximum | ReLU off)
        elif pooled:            # exact-f32 mode: one kernel writes the kept pre-pool activation and the pooled one
            a, nxt = ops.conv3x3_train(cur, packed[layer - 2], c.bias.detach(), cout)
        else:
            a = nxt = ops.conv3x3(cur, packed[layer - 2], c.bias.detach(), cout, pool=False, act=True)
        tape["layers"][layer] = (cur, a)
        cur = nxt
    return cur, tape


def fc_forward(emb, h, dtype):
    """Training forward of the three Linear + ReLU layers (vggish.py:13-19): h (N, 12288) in `dtype` -> ((N, 128) float32, tape)."""
    fcs = emb._fcs
    if dtype == torch.float32:
        ws = [f.weight.detach() for f in fcs]
    else:
        ws = emb._cache.get([f.weight for f in fcs], dtype, lambda: [ops.to_bf16(f.weight.detach().contiguous()) for f in fcs])
    tape = {"fc": [], "fc_w": ws, "dtype": dtype}
    for i, (f, w) in enumerate(zip(fcs, ws)):
        last = i == len(fcs) - 1
        out = ops.linear(h, w, f.bias.detach(), relu=True, out_dtype=torch.float32 if last else dtype)
        tape["fc"].append((h, out))
        h = out
    return h, tape


def forward(cnn_model, x, precision="f32"):
    """x: (N, 1, 96, 64) or (N, 96, 64) float32 -> ((N, 128) float32 embeddings or (N, 12288) bottlenecks, tape)."""
    feats, fcs, _, _ = _parts(cnn_model)
    dtype = torch.bfloat16 if precision == "bf16" else torch.float32
    cur, tape = features_forward(feats, x, dtype)
    h = cur.reshape(cur.shape[0], -1)
    tape["fc"] = []
    if fcs:
        h, t_fc = fc_forward(cnn_model.embeddings, h, dtype)
        tape["fc"], tape["fc_w"] = t_fc["fc"], t_fc["fc_w"]
    elif dtype != torch.float32:
        h = ops.to_f32(h)                       # just_bottlenecks: the head takes float32 features
    return h, tape


def _linear_backward_bf16(x_in, w_bf16, dz, g_w, g_b, want_dx):
    """bf16 form of mla_train._linear_backward: x_in (M, Kin), dz (M, N), w_bf16 (N, Kin), all bf16; g_w / g_b f32 (or None)."""
    if g_w is not None:
        ops.linear(ops.transpose_padded(dz), ops.transpose_padded(x_in), None, out=g_w, out_dtype=torch.float32)
    if g_b is not None:
        ops.col_sum(dz, g_b)
    if not want_dx:
        return None
    return ops.linear(dz, ops.transpose_padded(w_bf16), None, out_dtype=torch.bfloat16)


CONV_IDX = [0, 3, 6, 8, 11, 13]         # Sequential indices of the conv layers (vggish.py:108-118) and of the Linear layers (:13-19)
FC_IDX = [0, 2, 4]


def fc_backward(fcs, tape, d, grads, keys, want_dx, done=None):
    """Backward of fc_forward. d: gradient of its float32 result. grads[keys[i] + "weight" | "bias"]: float32 tensors to fill (a
    missing entry = that gradient is not wanted). Stops at the lowest Linear layer that has an entry unless `want_dx` (something
    below the Linear layers needs the input gradient, which is then returned: (N, 12288) in the tape's dtype)."""
    lp = tape["dtype"] == torch.bfloat16
    wanted = [k + "weight" in grads or k + "bias" in grads for k in keys]
    lowest = 0 if want_dx else (wanted.index(True) if any(wanted) else len(keys))
    for i in range(len(fcs) - 1, lowest - 1, -1):
        h_in, h_out = tape["fc"][i]
        dz = ops.relu_pool_bwd(h_out, d, pool=False, bf16=lp)
        dx_needed = want_dx or i > lowest
        if lp:
            d = _linear_backward_bf16(h_in, tape["fc_w"][i], dz, grads.get(keys[i] + "weight"), grads.get(keys[i] + "bias"), dx_needed)
        else:
            d = mla_train._linear_backward(h_in, fcs[i].weight.detach(), dz, grads.get(keys[i] + "weight"), grads.get(keys[i] + "bias"), dx_needed)
        if done:
            done(6 + i)
    return d if want_dx else None


def features_backward(feats, tape, d, grads, keys, done=None):
    """Backward of features_forward. d: gradient of its (N, 6, 4, 512) result in the tape's dtype. Fills grads[keys[pos] + "weight" |
    "bias"] (pos 0..5 = conv1..conv6) where present; the chain of input gradients stops at the lowest layer that has an entry, and
    conv1's backward (which recomputes the layer) runs only if conv1 itself has one."""
    convs = feats._convs
    wanted = [k + "weight" in grads or k + "bias" in grads for k in keys]
    if not any(wanted):
        return
    lowest = wanted.index(True)
    g = grads.get
    n = tape["x"].shape[0]
    d = d.reshape(n, 6, 4, 512)
    for layer in range(6, 1, -1):
        pos = layer - 1
        if pos < lowest:
            return
        cin, cout, H, W_, pooled = GEOM[layer]
        a_in, a = tape["layers"][layer]
        key = keys[pos]
        if a.dtype == torch.uint8:                                           # window codes of the training forward (bf16 mode)
            dz = ops.pool_bwd_codes(a, d.contiguous(), db=g(key + "bias"))
        else:
            dz = ops.relu_pool_bwd(a, d.contiguous(), pool=pooled, db=g(key + "bias"))     # bias gradient summed on the way
        if g(key + "weight") is not None:
            ops.conv_wgrad(dz, a_in, g(key + "weight"))
        if done:
            done(pos)
        if pos > lowest:
            wd = ops.repack_dgrad(convs[layer - 1].weight.detach().contiguous(), tape["dtype"])
            d = ops.conv3x3(dz, wd, None, cin, pool=False, act=False)
    if lowest > 0:          # conv1 is not in the update set: `d` is still the gradient of the lowest trained layer's OUTPUT, not conv1's
        return
    key = keys[0]
    dw = g(key + "weight") if g(key + "weight") is not None else torch.empty((64, 1, 3, 3), dtype=torch.float32, device=d.device)
    db = g(key + "bias") if g(key + "bias") is not None else torch.empty(64, dtype=torch.float32, device=d.device)
    ops.conv1_bwd(tape["x"], convs[0].weight.detach().contiguous(), convs[0].bias.detach(), d.contiguous(), dw, db)
    if done:
        done(0)


def backward(cnn_model, tape, d_out, grads, prefix, after_layer=None):
    """d_out: gradient w.r.t. forward()'s result (float32). Fills grads[prefix + <state_dict key>] for every CNN parameter that
    HAS an entry in `grads` (float32, weights in state_dict layout). Parameters without an entry get no gradient (they are
    frozen or not held by the caller's optimizer), and the chain of input gradients stops at the lowest layer that has one.
    after_layer(pos): called when the gradients of layer `pos` (0..5 conv, 6..8 Linear) are complete -- the hook the
    data-parallel step uses to start reducing finished gradient buckets while the layers below are still running."""
    feats, fcs, kf, ke = _parts(cnn_model)
    conv_keys = [prefix + kf + "%d." % i for i in CONV_IDX]
    fc_keys = [prefix + ke + "%d." % i for i in FC_IDX[:len(fcs)]] if fcs else []
    conv_wanted = any(k + "weight" in grads or k + "bias" in grads for k in conv_keys)
    d = d_out.contiguous()
    if fcs:
        d = fc_backward(fcs, tape, d, grads, fc_keys, conv_wanted, after_layer)
    elif tape["dtype"] == torch.bfloat16:
        d = ops.to_bf16(d)
    if conv_wanted:
        features_backward(feats, tape, d, grads, conv_keys, after_layer)
