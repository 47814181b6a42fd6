"""``torch.autograd.Function`` wrappers that make the drop-in modules differentiable, so that the reference's literal
training loop (train.py:124-138)

    optimizer.zero_grad(); outputs = clf(inputs); loss = criterion(outputs, labels); loss.backward(); optimizer.step()

runs unchanged on the drop-in ``Ensemble`` with any torch optimizer: the forward of each wrapped stage is the HIP training
forward (activations kept on a tape), its backward the HIP backward kernels of ``cnn_train`` / ``mla_train``; the gradients
come back to autograd as the gradients of the stage's parameters, which are inputs of the Function, so ``p.grad``
accumulates exactly as for a torch module and stays ``None`` for parameters that got no gradient (the dead
``attention_modules.*.fcf``: model.py:231 vs :237-238) or do not require one.

Three stages, one Function each, chained by ordinary autograd (reshapes / transposes between them are torch views):

* ``FeaturesFn``   -- the conv stack (vggish.py:108-118): ``VGGFeatures.forward``
* ``EmbeddingsFn`` -- the three Linear + ReLU layers (vggish.py:13-19): ``VGGEmbeddings.forward``
* ``HeadFn``       -- the whole multi-level-attention head (model.py:258-269): ``MultiLevelAttention.forward``

``TrainStep`` (train.py of this package) stays the fast path: one flat gradient buffer, fused Adam, SyncBN and bucketed
all-reduce. This module trades that for the reference's calling convention.

Not reproduced: the gradient with respect to the input examples (``waveform_to_examples`` returns a tensor with
``requires_grad=True``, vggish_input.py:79-80; the reference computes that gradient and never reads it) -- the conv stack's
Function returns ``None`` for it. Train and eval mode are both differentiable (eval: BatchNorm with its running statistics).
"""

import torch

from . import cnn_train, mla_train, ops

_DTYPES = {"f32": torch.float32, "bf16": torch.bfloat16}


def grad_precision(precision):
    if precision not in _DTYPES:
        raise NotImplementedError("CNN gradients are built for precision 'f32' (exact) and 'bf16' (bf16 arithmetic, f32 master "
                                  "weights), not %r" % (precision,))
    return _DTYPES[precision]


def _wanted(ctx, first, names):
    """{name: fresh float32 gradient tensor} for the parameter inputs (positions first, first + 1, ...) autograd asks for."""
    return {n: torch.empty(p_shape, dtype=torch.float32, device=ctx.device)
            for i, (n, p_shape) in enumerate(names) if ctx.needs_input_grad[first + i]}


class FeaturesFn(torch.autograd.Function):
    """apply(feats, dtype, x, *[w0, b0, w3, b3, ...]) -> (N, 6, 4, 512) NHWC activations in `dtype`."""

    @staticmethod
    def forward(ctx, feats, dtype, x, *params):
        out, tape = cnn_train.features_forward(feats, x, dtype)
        ctx.feats, ctx.tape, ctx.device = feats, tape, out.device
        ctx.names = [("%d.%s" % (i, leaf), tuple(getattr(feats[i], leaf).shape)) for i in cnn_train.CONV_IDX for leaf in ("weight", "bias")]
        return out

    @staticmethod
    def backward(ctx, d_out):
        grads = _wanted(ctx, 3, ctx.names)
        tape, ctx.tape = ctx.tape, None                           # the kept activations are released with this call
        if tape is None:
            raise RuntimeError("the conv stack's tape has been consumed: backward through it a second time needs a second forward")
        d = d_out.contiguous()
        if d.dtype != tape["dtype"]:
            d = ops.to_bf16(d.float().contiguous()) if tape["dtype"] == torch.bfloat16 else ops.to_f32(d)
        cnn_train.features_backward(ctx.feats, tape, d, grads, ["%d." % i for i in cnn_train.CONV_IDX])
        return (None, None, None) + tuple(grads.get(n) for n, _ in ctx.names)


class EmbeddingsFn(torch.autograd.Function):
    """apply(emb, dtype, h, *[w0, b0, w2, b2, w4, b4]): h (N, 12288) in `dtype` -> (N, 128) float32."""

    @staticmethod
    def forward(ctx, emb, dtype, h, *params):
        out, tape = cnn_train.fc_forward(emb, h.detach(), dtype)
        ctx.emb, ctx.tape, ctx.device = emb, tape, out.device
        ctx.names = [("%d.%s" % (i, leaf), tuple(getattr(emb[i], leaf).shape)) for i in cnn_train.FC_IDX for leaf in ("weight", "bias")]
        return out

    @staticmethod
    def backward(ctx, d_out):
        grads = _wanted(ctx, 3, ctx.names)
        tape, ctx.tape = ctx.tape, None
        if tape is None:
            raise RuntimeError("the embeddings' tape has been consumed: backward through it a second time needs a second forward")
        d = cnn_train.fc_backward(ctx.emb._fcs, tape, d_out.float().contiguous(), grads, ["%d." % i for i in cnn_train.FC_IDX],
                                  want_dx=ctx.needs_input_grad[2])
        return (None, None, d) + tuple(grads.get(n) for n, _ in ctx.names)


class CastFn(torch.autograd.Function):
    """bf16 bottlenecks (just_bottlenecks=True) -> the float32 features the head takes; the gradient goes back in bf16."""

    @staticmethod
    def forward(ctx, x):
        return ops.to_f32(x.detach().contiguous())

    @staticmethod
    def backward(ctx, d):
        return ops.to_bf16(d.float().contiguous())


class HeadFn(torch.autograd.Function):
    """apply(mla, x, *parameters of mla in named_parameters() order): x (B, T, M) float32 -> (B, K) sigmoid scores, in the module's
    current mode: train (BatchNorm batch statistics, running buffers updated, Dropout) or eval (running statistics, no dropout);
    the backward differentiates the forward that ran."""

    @staticmethod
    def forward(ctx, mla, x, *params):
        c = mla_train.Ctx(tape=True)
        out = mla_train.mla_forward(mla, x.detach(), c)
        ctx.mla, ctx.c, ctx.device, ctx.x_shape = mla, c, out.device, tuple(x.shape)
        ctx.names = [(n, tuple(p.shape)) for n, p in mla.named_parameters()]
        return out

    @staticmethod
    def backward(ctx, d_out):
        c, ctx.c = ctx.c, None
        if c is None:
            raise RuntimeError("the head's tape has been consumed: backward through it a second time needs a second forward")
        # the head's backward always runs whole (0.3 ms); gradients nobody asked for land in scratch tensors
        full = {n: torch.empty(s, dtype=torch.float32, device=ctx.device) for n, s in ctx.names if ".fcf." not in n}
        dx = mla_train.mla_backward(ctx.mla, c, d_out.float().contiguous(), full, need_input_grad=ctx.needs_input_grad[1])
        if dx is not None:
            dx = dx.reshape(ctx.x_shape)
        return (None, dx) + tuple(full.get(n) if ctx.needs_input_grad[2 + i] else None for i, (n, _) in enumerate(ctx.names))


def wants_grad(module, *inputs):
    """Does a call of `module` have to be recorded for autograd? Gradients are enabled and a parameter (or an input) requires one."""
    if not torch.is_grad_enabled():
        return False
    return any(torch.is_tensor(t) and t.requires_grad for t in inputs) or any(p.requires_grad for p in module.parameters())
