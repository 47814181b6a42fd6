"""Portable seeded tensors: the same bits on every host, numpy version and device.

Neither side of a parity test may depend on ``torch.manual_seed`` (its stream
differs between CPU and GPU generators) and the 73 M-parameter VGGish weights are
too large to commit, so the golden-fixture generator, the oracle, the tests and
``bench.py`` all regenerate weights, waveforms and dropout masks from this
counter-based integer hash (splitmix64 finaliser over ``(seed, stream, index)``).
Only uint64 wrap-around arithmetic is used, so results are bit-identical
everywhere.

The state_dict key set generated here is the reference's
(``model.py:200-269`` / ``vggish.py:9-31,108-118`` module attribute names), see
SURVEY.md section 5 "Checkpoint / resume".
"""

import math
import zlib

import numpy as np

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _mix(z):
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def stream_id(name):
    """Stable 32-bit id of a tensor name (crc32 is specified bit-exactly)."""
    return zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF


def bits24(seed, stream, n, offset=0):
    """n 24-bit integers for (seed, stream), starting at element ``offset``."""
    with np.errstate(over="ignore"):
        key = _mix(np.uint64(seed) * _GOLD + np.uint64(stream))
        idx = np.arange(offset, offset + n, dtype=np.uint64)
        z = _mix(idx * _GOLD + key)
    return (z >> np.uint64(40)).astype(np.int64)


def uniform(seed, stream, n, lo=-1.0, hi=1.0, dtype=np.float32, offset=0):
    """Uniform [lo, hi) values; the 24-bit draw is exact in float32."""
    u = bits24(seed, stream, n, offset).astype(np.float64) / float(1 << 24)
    return (lo + (hi - lo) * u).astype(dtype)


def keep_mask(seed, stream, n, p_drop):
    """Dropout keep-mask (1 = keep) with P(drop) = p_drop, as uint8."""
    thresh = int(round(p_drop * (1 << 24)))
    return (bits24(seed, stream, n) >= thresh).astype(np.uint8)


def waveform(seed, n_samples, n_waveforms=1, dtype=np.float32):
    """Synthetic 16 kHz mono PCM in [-1, 1): (n_waveforms, n_samples)."""
    out = np.empty((n_waveforms, n_samples), dtype=dtype)
    for w in range(n_waveforms):
        out[w] = uniform(seed, stream_id("pcm/%d" % w), n_samples, dtype=dtype)
    return out


# ---------------------------------------------------------------------------
# state_dict layout of the reference model (vggish branch), model.py / vggish.py
# ---------------------------------------------------------------------------

VGG_CONV = [(0, 1, 64), (3, 64, 128), (6, 128, 256), (8, 256, 256), (11, 256, 512), (13, 512, 512)]
VGG_FC = [(0, 12288, 4096), (2, 4096, 4096), (4, 4096, 128)]


def vggish_shapes(prefix=""):
    """Ordered {key: shape} for VGG.features / VGG.embeddings (vggish.py:13-19, 108-118)."""
    shapes = {}
    for idx, cin, cout in VGG_CONV:
        shapes["%sfeatures.%d.weight" % (prefix, idx)] = (cout, cin, 3, 3)
        shapes["%sfeatures.%d.bias" % (prefix, idx)] = (cout,)
    for idx, fin, fout in VGG_FC:
        shapes["%sembeddings.%d.weight" % (prefix, idx)] = (fout, fin)
        shapes["%sembeddings.%d.bias" % (prefix, idx)] = (fout,)
    return shapes


def _bn_shapes(shapes, key, c):
    shapes[key + ".weight"] = (c,)
    shapes[key + ".bias"] = (c,)
    shapes[key + ".running_mean"] = (c,)
    shapes[key + ".running_var"] = (c,)
    shapes[key + ".num_batches_tracked"] = ()


def mla_shapes(model_conf, emb_input_size, prefix="mla.", T=10, H=600, K=10):
    """Ordered {key: shape} for MultiLevelAttention (model.py:200-269)."""
    shapes = {}
    for lvl, n_fc in enumerate(model_conf):
        p = "%sembedded_mappings.%d." % (prefix, lvl)
        _bn_shapes(shapes, p + "norm0", T)
        for j in range(n_fc):
            fin = emb_input_size if (lvl == 0 and j == 0) else H
            shapes[p + "fc.%d.weight" % j] = (H, fin)
            shapes[p + "fc.%d.bias" % j] = (H,)
        for j in range(n_fc):
            _bn_shapes(shapes, p + "norms.%d" % j, T)
    for lvl in range(len(model_conf)):
        p = "%sattention_modules.%d." % (prefix, lvl)
        for nm in ("fcv", "fcf"):
            shapes[p + nm + ".weight"] = (K, H)
            shapes[p + nm + ".bias"] = (K,)
        _bn_shapes(shapes, p + "normv", T)
        _bn_shapes(shapes, p + "normf", T)
    shapes[prefix + "fc.weight"] = (K, len(model_conf) * K)
    shapes[prefix + "fc.bias"] = (K,)
    _bn_shapes(shapes, prefix + "norm", K)
    return shapes


def ensemble_shapes(model_conf=(2, 1), just_bottlenecks=False):
    """Ordered {key: shape} of Ensemble.state_dict() for cnn_type='vggish' (model.py:54-56)."""
    emb = 12288 if just_bottlenecks else 128
    shapes = mla_shapes(list(model_conf), emb)
    if just_bottlenecks:
        # CNN.cnn_model = Sequential(features, CnnFlatten): keys cnn.cnn_model.0.<idx>.*
        for idx, cin, cout in VGG_CONV:
            shapes["cnn.cnn_model.0.%d.weight" % idx] = (cout, cin, 3, 3)
            shapes["cnn.cnn_model.0.%d.bias" % idx] = (cout,)
    else:
        shapes.update(vggish_shapes("cnn.cnn_model."))
    return shapes


def make_tensor(seed, key, shape):
    """One synthetic tensor for a state_dict key (float32, or int64 for counters).

    Conv / Linear weights use a Kaiming-uniform bound sqrt(6 / fan_in) so that
    activations keep O(1) scale through the ReLU stack (a parity test on vanishing
    activations would not see kernel errors); biases use 1/sqrt(fan_in).
    BatchNorm affine / running statistics get non-trivial values so that
    eval-mode normalisation is exercised.
    """
    sid = stream_id(key)
    n = int(np.prod(shape)) if len(shape) else 1
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return np.zeros((), dtype=np.int64)
    u = uniform(seed, sid, n)
    if leaf == "running_mean":
        v = 0.1 * u
    elif leaf == "running_var":
        v = 1.0 + 0.5 * u
    elif len(shape) == 1 and ("norm" in key):
        v = (1.0 + 0.1 * u) if leaf == "weight" else 0.1 * u
    elif leaf == "weight":
        fan_in = int(np.prod(shape[1:]))
        v = math.sqrt(6.0 / fan_in) * u
    else:  # Linear / Conv bias
        v = 0.1 * u
    return np.asarray(v, dtype=np.float32).reshape(shape)


def make_state_dict(seed, shapes):
    return {k: make_tensor(seed, k, s) for k, s in shapes.items()}
