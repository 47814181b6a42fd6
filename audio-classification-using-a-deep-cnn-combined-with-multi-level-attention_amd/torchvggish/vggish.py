"""Drop-in for the reference's ``torchvggish/vggish.py``: ``VGG``, ``Postprocessor``,
``make_layers`` and ``VGGish`` with the same constructor arguments, attribute names and
``state_dict`` keys (``features.{0,3,6,8,11,13}.*``, ``embeddings.{0,2,4}.*``), computed by the
HIP kernels of libmla_hip.so (csrc/conv.hip, csrc/gemm.hip).

The modules inside ``features`` / ``embeddings`` only HOLD parameters in the reference's
layout; the containers run the fused HIP pipeline (conv + bias + ReLU + max-pool in one
kernel per conv, NHWC activations). Extra keyword ``precision`` ("f32" exact-MFMA parity
mode, default, "bf16", or "bf16x3": f32-grade results from three bf16 MFMA products per term) is the one addition to
the reference signatures.
"""

import math

import numpy as np
import torch
import torch.nn as nn

from . import vggish_input, vggish_params
from .. import ops

_DTYPES = {"f32": torch.float32, "bf16": torch.bfloat16, "bf16x3": torch.bfloat16}     # bf16x3: split planes, see MLA_BF16X3


class Conv3x3(nn.Module):
    """Parameter holder with nn.Conv2d(cin, cout, 3, padding=1)'s parameter names, shapes and
    default initialisation (vggish.py:113)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, 3, 3))
        self.bias = nn.Parameter(torch.empty(out_channels))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(in_channels * 9)
        nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x):
        raise RuntimeError("Conv3x3 is executed by its VGGFeatures container (fused HIP pipeline)")


class _Marker(nn.Module):
    def forward(self, x):
        raise RuntimeError("%s is fused into the preceding HIP convolution" % type(self).__name__)


class ReLU(_Marker):
    pass


class MaxPool2x2(_Marker):
    pass


class _Cache:
    """Derived (repacked / bf16) weight copies, refreshed when a parameter changes in place."""

    def __init__(self):
        self.key, self.val = None, None

    def get(self, params, dtype, make):
        key = (dtype,) + tuple((p.data_ptr(), p._version) for p in params)
        if key != self.key:
            self.key, self.val = key, make()
        return self.val


class VGGFeatures(nn.Sequential):
    """nn.Sequential with the reference's layer indices (vggish.py:108-118) whose forward is the
    HIP conv stack: (N, 1, 96, 64) -> (N, 512, 6, 4). The result is an NCHW-shaped VIEW of NHWC
    memory, so the reference's transpose/contiguous flatten (vggish.py:26-29) costs nothing."""

    precision = "f32"

    def __init__(self, *layers):
        super().__init__(*layers)
        self._cache = _Cache()
        convs = [m for m in self if isinstance(m, Conv3x3)]
        assert [(c.in_channels, c.out_channels) for c in convs] == [(1, 64), (64, 128), (128, 256), (256, 256), (256, 512), (512, 512)]
        self._convs = convs

    def forward_nhwc(self, x, dtype):
        """x: (N, 96, 64) or (N, 1, 96, 64) examples, f32 or bf16 -> (N, 6, 4, 512) NHWC
        (precision "bf16x3": (N, 6, 4, 1024) = [hi(512) | lo(512)] per pixel)."""
        x = x.detach().reshape(-1, 96, 64)
        if x.dtype not in (torch.float32, torch.bfloat16):
            x = x.float()
        x = x.contiguous()
        convs = self._convs
        if self.precision == "bf16x3":
            packed = self._cache.get([c.weight for c in convs[1:]], "bf16x3",
                                     lambda: [ops.split_conv_weight(c.weight.detach().contiguous()) for c in convs[1:]])
            h = ops.conv1(x.float(), convs[0].weight.detach().contiguous(), convs[0].bias.detach(), torch.bfloat16, split=True)
            for layer, (c, w) in enumerate(zip(convs[1:], packed), start=2):
                h = ops.conv(layer, h, w, c.bias.detach(), split=True)
            return h
        packed = self._cache.get([c.weight for c in convs[1:]], dtype,
                                 lambda: [ops.repack_conv_weight(c.weight.detach().contiguous(), dtype) for c in convs[1:]])
        h = ops.conv1(x, convs[0].weight.detach().contiguous(), convs[0].bias.detach(), dtype)
        for layer, (c, w) in enumerate(zip(convs[1:], packed), start=2):
            h = ops.conv(layer, h, w, c.bias.detach())
        return h

    def forward(self, x):
        from .. import differentiable as D
        if D.wants_grad(self):
            # a conv parameter requires grad and autograd is recording (the reference's loop, train.py:124-138, after
            # cnn_trainable=True / set_requires_grad): training forward with a tape, HIP backward kernels behind loss.backward()
            params = [t for i in (0, 3, 6, 8, 11, 13) for t in (self[i].weight, self[i].bias)]
            return D.FeaturesFn.apply(self, D.grad_precision(self.precision), x, *params).permute(0, 3, 1, 2)
        return self.forward_nhwc(x, _DTYPES[self.precision]).permute(0, 3, 1, 2)


class Linear(nn.Module):
    """Parameter holder with nn.Linear's names, shapes and default initialisation."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(in_features)
        nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x):
        raise RuntimeError("Linear is executed by its container (HIP GEMM)")


class VGGEmbeddings(nn.Sequential):
    """vggish.py:13-19: Linear(12288, 4096)+ReLU, Linear(4096, 4096)+ReLU, Linear(4096, 128)+ReLU
    as three MFMA GEMMs with fused bias + ReLU; the last one writes float32 embeddings."""

    precision = "f32"

    def __init__(self, *layers):
        super().__init__(*layers)
        self._cache = _Cache()
        self._fcs = [m for m in self if isinstance(m, Linear)]

    def forward(self, x):
        dtype = _DTYPES[self.precision]
        fcs = self._fcs
        from .. import differentiable as D
        if D.wants_grad(self, x):
            assert x.dtype == D.grad_precision(self.precision) and x.is_contiguous(), "embeddings input must come from VGGFeatures in the same precision"
            return D.EmbeddingsFn.apply(self, dtype, x, *[t for f in fcs for t in (f.weight, f.bias)])
        if self.precision == "bf16x3":
            # the conv stack's flatten is pixel-major with [hi(512) | lo(512)] per pixel; later layers have one plane pair
            segs = [512, fcs[1].in_features, fcs[2].in_features]
            ws = self._cache.get([f.weight for f in fcs], "bf16x3",
                                 lambda: [ops.split_linear_weight(f.weight.detach().contiguous(), sg) for f, sg in zip(fcs, segs)])
            h = x.detach()
            assert h.dtype == torch.bfloat16 and h.is_contiguous() and h.shape[1] == 2 * fcs[0].in_features
            for i, (f, w, sg) in enumerate(zip(fcs, ws, segs)):
                h = ops.linear_split(h, w, f.bias.detach(), sg, relu=True, out_split=i < len(fcs) - 1)
            return h
        if dtype == torch.float32:
            ws = [f.weight.detach() for f in fcs]
        else:
            ws = self._cache.get([f.weight for f in fcs], dtype, lambda: [ops.to_bf16(f.weight.detach().contiguous()) for f in fcs])
        h = x.detach()
        assert h.dtype == dtype and h.is_contiguous(), "embeddings input must come from VGGFeatures in the same precision"
        for i, (f, w) in enumerate(zip(fcs, ws)):
            last = i == len(fcs) - 1
            h = ops.linear(h, w, f.bias.detach(), relu=True, out_dtype=torch.float32 if last else dtype)
        return h


def make_layers():
    """[64, M, 128, M, 256, 256, M, 512, 512, M] (vggish.py:108-118), same Sequential indices."""
    layers, in_channels = [], 1
    for v in [64, "M", 128, "M", 256, 256, "M", 512, 512, "M"]:
        if v == "M":
            layers.append(MaxPool2x2())
        else:
            layers += [Conv3x3(in_channels, v), ReLU()]
            in_channels = v
    return VGGFeatures(*layers)


class VGG(nn.Module):
    def __init__(self, features):
        super().__init__()
        self.features = features
        self.embeddings = VGGEmbeddings(Linear(512 * 4 * 6, 4096), ReLU(), Linear(4096, 4096), ReLU(),
                                        Linear(4096, 128), ReLU())

    def set_precision(self, precision):
        assert precision in _DTYPES
        self.features.precision = precision
        self.embeddings.precision = precision
        return self

    def forward(self, x):
        x = self.features(x)
        # vggish.py:26-29: NCHW -> (N, h*w*c). `x` is an NCHW view of NHWC memory: no data moves.
        x = torch.transpose(x, 1, 3)
        x = torch.transpose(x, 1, 2)
        x = x.contiguous()
        x = x.view(x.size(0), -1)
        return self.embeddings(x)


class Postprocessor(nn.Module):
    """PCA + clamp + 8-bit quantisation of the embeddings (vggish.py:34-105); parameters
    ``pca_eigen_vectors`` (128, 128) and ``pca_means`` (128, 1) as in the reference."""

    def __init__(self):
        super().__init__()
        n = vggish_params.EMBEDDING_SIZE
        self.pca_eigen_vectors = nn.Parameter(torch.empty((n, n), dtype=torch.float), requires_grad=False)
        self.pca_means = nn.Parameter(torch.empty((n, 1), dtype=torch.float), requires_grad=False)

    def postprocess(self, embeddings_batch):
        assert len(embeddings_batch.shape) == 2, "Expected 2-d batch, got %r" % (embeddings_batch.shape,)
        assert embeddings_batch.shape[1] == vggish_params.EMBEDDING_SIZE, "Bad batch shape: %r" % (embeddings_batch.shape,)
        import ctypes
        from .. import _lib
        x = embeddings_batch.detach().float().contiguous()
        y = torch.empty_like(x)
        vp = ctypes.c_void_p
        _lib.check(_lib.lib().mla_postprocess(vp(x.data_ptr()), vp(self.pca_eigen_vectors.detach().contiguous().data_ptr()),
                                              vp(self.pca_means.detach().contiguous().data_ptr()), x.shape[0], vp(y.data_ptr()),
                                              _lib.stream_ptr()))
        return torch.squeeze(y)

    def forward(self, x):
        return self.postprocess(x)


class VGGish(VGG):
    """vggish.py:143-184. ``pretrained=True`` needs a network fetch (torch.hub) and is refused
    here; load weights with ``load_state_dict`` (keys are the reference's)."""

    def __init__(self, urls, pretrained=True, preprocess=True, postprocess=True, progress=True, precision="f32"):
        super().__init__(make_layers())
        if pretrained:
            raise RuntimeError("pretrained=True downloads %r; there is no network here -- construct with "
                               "pretrained=False and load_state_dict() the checkpoint" % (urls,))
        self.preprocess = preprocess
        self.postprocess = postprocess
        if self.postprocess:
            self.pproc = Postprocessor()
        self.set_precision(precision)

    def forward(self, x, fs=None):
        if self.preprocess:
            x = self._preprocess(x, fs)
        x = VGG.forward(self, x)
        if self.postprocess:
            x = self._postprocess(x)
        return x

    def _preprocess(self, x, fs):
        if isinstance(x, np.ndarray):
            x = vggish_input.waveform_to_examples(x, fs)
        elif isinstance(x, str):
            x = vggish_input.wavfile_to_examples(x)
        else:
            raise AttributeError
        return x

    def _postprocess(self, x):
        return self.pproc(x)
