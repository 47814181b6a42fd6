"""Tensor-level wrappers over the C ABI (include/mla_hip.h). PyTorch supplies device
buffers and the current stream; every number is produced by a HIP kernel of libmla_hip.so.
Nothing here falls back to torch ops."""

import ctypes

import torch

from . import _lib

DT = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16, torch.int16: _lib.I16}
BN_EPS = 1e-5

# Optional per-kernel timing (bench.py): when `profile` is a list, every wrapped call appends
# (name, start_event, end_event) recorded on the current stream -- the stream the kernels run on.
profile = None


def _timed(name, fn, *args):
    if profile is None:
        return fn(*args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = fn(*args)
    e1.record()
    profile.append((name, e0, e1))
    return rc


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _chk(t, dtype=None):
    assert t.is_cuda and t.is_contiguous(), "HIP ops need contiguous CUDA tensors"
    if dtype is not None:
        assert t.dtype == dtype, (t.dtype, dtype)
    return t


def repack_conv_weight(w, dtype):
    """(Cout, Cin, 3, 3) f32 -> (Cout, 9, Cin) in `dtype`."""
    _chk(w, torch.float32)
    cout, cin = w.shape[0], w.shape[1]
    out = torch.empty((cout, 9, cin), dtype=dtype, device=w.device)
    _lib.check(_lib.lib().mla_conv_repack_weights(_p(w), cout, cin, _p(out), DT[dtype], _lib.stream_ptr()))
    return out


def to_bf16(w):
    _chk(w, torch.float32)
    out = torch.empty(w.shape, dtype=torch.bfloat16, device=w.device)
    _lib.check(_lib.lib().mla_convert_f32(_p(w), _p(out), w.numel(), _lib.BF16, _lib.stream_ptr()))
    return out


def to_f32(x):
    _chk(x, torch.bfloat16)
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mla_convert_bf16_to_f32(_p(x), _p(out), x.numel(), _lib.stream_ptr()))
    return out


def conv1(x, w, b, dtype):
    """x (N, 96, 64) f32|bf16 -> (N, 48, 32, 64) NHWC."""
    _chk(x); _chk(w, torch.float32); _chk(b, torch.float32)
    n = x.shape[0]
    assert tuple(x.shape[1:]) == (96, 64) and tuple(w.shape) == (64, 1, 3, 3)
    out = torch.empty((n, 48, 32, 64), dtype=dtype, device=x.device)
    _lib.check(_timed("conv1", _lib.lib().mla_vggish_conv1, _p(x), DT[x.dtype], n, _p(w), _p(b), _p(out), DT[dtype], _lib.stream_ptr()))
    return out


CONV_SHAPES = {2: ((48, 32, 64), (24, 16, 128)), 3: ((24, 16, 128), (24, 16, 256)), 4: ((24, 16, 256), (12, 8, 256)),
               5: ((12, 8, 256), (12, 8, 512)), 6: ((12, 8, 512), (6, 4, 512))}


def conv(layer, x, w_packed, b):
    """VGGish conv `layer` (2..6) with fused bias + ReLU (+ 2x2 max-pool for 2, 4, 6). NHWC."""
    _chk(x); _chk(w_packed, x.dtype); _chk(b, torch.float32)
    shp_in, shp_out = CONV_SHAPES[layer]
    assert tuple(x.shape[1:]) == shp_in, (x.shape, shp_in)
    assert tuple(w_packed.shape) == (shp_out[2], 9, shp_in[2])
    n = x.shape[0]
    out = torch.empty((n,) + shp_out, dtype=x.dtype, device=x.device)
    _lib.check(_timed("conv%d" % layer, _lib.lib().mla_vggish_conv, layer, _p(x), _p(w_packed), _p(b), _p(out), n, DT[x.dtype], _lib.stream_ptr()))
    return out


def linear(a, w, b, relu=False, out_dtype=None):
    """a (M, K), w (N, K) same dtype (f32 | bf16), bias f32 -> (M, N)."""
    _chk(a); _chk(w, a.dtype)
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K
    out_dtype = out_dtype or a.dtype
    out = torch.empty((M, N), dtype=out_dtype, device=a.device)
    _lib.check(_timed("linear_%dx%d" % (K, N), _lib.lib().mla_linear, _p(a), a.stride(0), _p(w), w.stride(0), _p(b), _p(out), N,
                      M, N, K, DT[a.dtype], DT[out_dtype], int(relu), _lib.stream_ptr()))
    return out


def linear_small(a, w, b):
    _chk(a, torch.float32); _chk(w, torch.float32)
    M, K = a.shape
    N = w.shape[0]
    out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    _lib.check(_lib.lib().mla_linear_small(_p(a), a.stride(0), _p(w), w.stride(0), _p(b), _p(out), N, M, N, K, _lib.stream_ptr()))
    return out


_ws = {}


def _workspace(device):
    key = str(device)
    if key not in _ws:
        _ws[key] = torch.empty(int(_lib.lib().mla_bn_stats_workspace_bytes()) // 8, dtype=torch.float64, device=device)
    return _ws[key]


def bn_stats(x, mode, period, running_mean=None, running_var=None, momentum=-1.0):
    """Batch mean / biased variance per channel (mode 0: channel = row % period; 1: column);
    optionally updates running statistics in place (unbiased variance, torch semantics)."""
    _chk(x, torch.float32)
    rows, cols = x.shape
    ch = period if mode == 0 else cols
    mean = torch.empty(ch, dtype=torch.float32, device=x.device)
    var = torch.empty(ch, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mla_bn_stats(_p(x), rows, cols, x.stride(0), mode, period, _p(_workspace(x.device)), _p(mean),
                                       _p(var), _p(running_mean), _p(running_var), float(momentum), _lib.stream_ptr()))
    return mean, var


def bn_apply(x, mode, period, mean, var, gamma, beta, act=0, keep_mask=None, drop_scale=1.0, out=None):
    _chk(x, torch.float32)
    rows, cols = x.shape
    if out is None:
        out = torch.empty_like(x)
    if keep_mask is not None:
        _chk(keep_mask, torch.uint8)
        assert keep_mask.numel() == rows * cols
    _lib.check(_lib.lib().mla_bn_apply(_p(x), x.stride(0), _p(out), out.stride(0), rows, cols, mode, period, _p(mean), _p(var),
                                       _p(gamma), _p(beta), BN_EPS, act, _p(keep_mask), float(drop_scale), _lib.stream_ptr()))
    return out


def attention_pool(z, bags, T, K, nv, nf, y, save=False):
    """z (bags*T, K); nv / nf = (mean, var, gamma, beta) of normv / normf; writes y (a (bags, K)
    column slice of the concatenated level outputs). Returns (att, cla) when save=True."""
    _chk(z, torch.float32)
    assert y.dtype == torch.float32 and y.stride(1) == 1
    att = torch.empty_like(z) if save else None
    cla = torch.empty_like(z) if save else None
    _lib.check(_lib.lib().mla_attention_pool(_p(z), bags, T, K, _p(nv[0]), _p(nv[1]), _p(nv[2]), _p(nv[3]), _p(nf[0]), _p(nf[1]),
                                             _p(nf[2]), _p(nf[3]), BN_EPS, _p(y), y.stride(0), _p(att), _p(cla), _lib.stream_ptr()))
    return att, cla
