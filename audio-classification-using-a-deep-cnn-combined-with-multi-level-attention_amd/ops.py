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
_ws = {}                  # per-device scratch buffers (statistics partials, column sums, split-K partials)


_event_pool = []


def reserve_events(n):
    """Create (and record once, which is what allocates them) n timing events ahead of a timed region: creating events inside
    it can stall the launching thread for tens of milliseconds when the runtime has to grow its event pool (measured: one
    60 ms stall in a 10-step region)."""
    while len(_event_pool) < n:
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        _event_pool.append(e)
    torch.cuda.synchronize()


def _event():
    return _event_pool.pop() if _event_pool else torch.cuda.Event(enable_timing=True)


def _timed(name, fn, *args):
    if profile is None:
        return fn(*args)
    e0, e1 = _event(), _event()
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


def split_conv_weight(w):
    """(Cout, Cin, 3, 3) f32 -> (Cout, 9, 3 Cin) bf16 = [hi | lo | hi] per 64-channel chunk and tap (bf16x3 mode)."""
    packed = repack_conv_weight(w, torch.float32)
    cout, _, cin = packed.shape
    out = torch.empty((cout, 9, 3 * cin), dtype=torch.bfloat16, device=w.device)
    _lib.check(_lib.lib().mla_split_bf16x3(_p(packed), cout * 9, cin, cin, _p(out), 3 * cin, 64, 3, _lib.stream_ptr()))
    return out


def split_linear_weight(w, seg):
    """(N, K) f32 -> (N, 3 K) bf16 = [hi | lo | hi] per segment of `seg` input features (bf16x3 mode)."""
    _chk(w, torch.float32)
    n, k = w.shape
    out = torch.empty((n, 3 * k), dtype=torch.bfloat16, device=w.device)
    _lib.check(_lib.lib().mla_split_bf16x3(_p(w), n, k, k, _p(out), 3 * k, seg, 3, _lib.stream_ptr()))
    return out


def merge_split(x, seg):
    """(rows, 2 cols) bf16 [hi | lo] per segment -> (rows, cols) f32."""
    _chk(x, torch.bfloat16)
    rows, c2 = x.shape
    out = torch.empty((rows, c2 // 2), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mla_merge_bf16x3(_p(x), rows, c2 // 2, c2, seg, _p(out), _lib.stream_ptr()))
    return out


def linear_split(a, w, b, seg, relu=True, out_split=True):
    """bf16x3 Linear: a (M, 2K) split activations, w (N, 3K) from split_linear_weight -> (M, 2N) split or (M, N) f32."""
    _chk(a, torch.bfloat16); _chk(w, torch.bfloat16)
    M, K = a.shape[0], a.shape[1] // 2
    N = w.shape[0]
    assert w.shape[1] == 3 * K
    out = torch.empty((M, 2 * N), dtype=torch.bfloat16, device=a.device) if out_split else torch.empty((M, N), dtype=torch.float32, device=a.device)
    _lib.check(_timed("linear_%dx%d" % (K, N), _lib.lib().mla_linear_bf16x3, _p(a), a.stride(0), _p(w), w.stride(0), _p(b), _p(out),
                      out.stride(0), M, N, K, seg, _lib.BF16X3 if out_split else _lib.F32, int(relu), _lib.stream_ptr()))
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


def conv1(x, w, b, dtype, split=False):
    """x (N, 96, 64) f32|bf16 -> (N, 48, 32, 64) NHWC; split (bf16x3): f32 in -> (N, 48, 32, 128) = [hi(64) | lo(64)]."""
    _chk(x); _chk(w, torch.float32); _chk(b, torch.float32)
    n = x.shape[0]
    assert tuple(x.shape[1:]) == (96, 64) and tuple(w.shape) == (64, 1, 3, 3)
    out = torch.empty((n, 48, 32, 128 if split else 64), dtype=dtype, device=x.device)
    _lib.check(_timed("conv1", _lib.lib().mla_vggish_conv1, _p(x), DT[x.dtype], n, _p(w), _p(b), _p(out),
                      _lib.BF16X3 if split else DT[dtype], _lib.stream_ptr()))
    return out


CONV_SHAPES = {2: ((48, 32, 64), (24, 16, 128)), 3: ((24, 16, 128), (24, 16, 256)), 4: ((24, 16, 256), (12, 8, 256)),
               5: ((12, 8, 256), (12, 8, 512)), 6: ((12, 8, 512), (6, 4, 512))}


def conv(layer, x, w_packed, b, split=False):
    """VGGish conv `layer` (2..6) with fused bias + ReLU (+ 2x2 max-pool for 2, 4, 6). NHWC. split (bf16x3): activations
    carry [hi | lo] planes (2 C channels), weights [hi | lo | hi] (3 Cin per tap)."""
    _chk(x); _chk(w_packed, x.dtype); _chk(b, torch.float32)
    shp_in, shp_out = CONV_SHAPES[layer]
    ka, kw, ko = (2, 3, 2) if split else (1, 1, 1)
    assert tuple(x.shape[1:]) == shp_in[:2] + (ka * shp_in[2],), (x.shape, shp_in)
    assert tuple(w_packed.shape) == (shp_out[2], 9, kw * shp_in[2])
    n = x.shape[0]
    out = torch.empty((n,) + shp_out[:2] + (ko * shp_out[2],), dtype=x.dtype, device=x.device)
    _lib.check(_timed("conv%d" % layer, _lib.lib().mla_vggish_conv, layer, _p(x), _p(w_packed), _p(b), _p(out), n,
                      _lib.BF16X3 if split else DT[x.dtype], _lib.stream_ptr()))
    return out


KSPLIT = 8          # fixed K ranges of the narrow forward layers (see linear)


def linear(a, w, b, relu=False, out_dtype=None, out=None, split_k=False):
    """a (M, K), w (N, K) same dtype (f32 | bf16), bias f32 -> (M, N) (optionally into `out`). split_k: allow K to be split
    over workgroups when there are few output tiles (weight gradients). Forward layers never split: the number of splits
    would depend on the batch size, and a bag's scores must not depend on which batch it is computed in."""
    _chk(a); _chk(w, a.dtype)
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K
    out_dtype = out_dtype or a.dtype
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=a.device)
    else:
        assert out.is_contiguous() and out.numel() == M * N and out.dtype == out_dtype
    # narrow outputs (the attention modules' 600 -> 10 fcv): a dedicated one-pass kernel instead of a nearly empty MFMA tile
    if (N <= 16 and a.dtype == torch.float32 and out_dtype == torch.float32 and not relu and not split_k and K % 4 == 0
            and K * N * 4 <= 65536 and a.stride(0) % 4 == 0 and w.stride(0) % 4 == 0
            and a.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0):
        _lib.check(_timed("linear_%dx%d" % (K, N), _lib.lib().mla_linear_narrow, _p(a), a.stride(0), _p(w), w.stride(0), _p(b), _p(out), N,
                          M, N, K, _lib.stream_ptr()))
        return out
    # narrow forward layers with a long reduction (VGGish's Linear(4096, 128): one column tile, so M / 128 workgroups for 256 CUs):
    # K always runs as KSPLIT fixed ranges whose partial sums are added in range order -- a per-LAYER constant, so a row's result
    # is the same in every batch (a split chosen by batch size would change the summation order)
    kc = 64 if a.dtype == torch.bfloat16 else 32
    if N <= 128 and K >= 2048 and K % (kc * KSPLIT) == 0 and not split_k and a.stride(0) % 8 == 0 and w.stride(0) % 8 == 0:
        key = "ksplit/" + str(a.device)
        if key not in _ws or _ws[key].numel() < KSPLIT * M * N:
            _ws[key] = torch.empty(max(KSPLIT * M * N, 1 << 22), dtype=torch.float32, device=a.device)
        ws = _ws[key]
        _lib.check(_timed("linear_%dx%d" % (K, N), _lib.lib().mla_linear_ksplit, _p(a), a.stride(0), _p(w), w.stride(0), _p(b), _p(out), N,
                          M, N, K, DT[a.dtype], DT[out_dtype], int(relu), KSPLIT, _p(ws), ws.numel(), _lib.stream_ptr()))
        return out
    # few output tiles but a long reduction (weight gradients): split K over workgroups
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    if split_k and a.dtype == torch.float32 and out_dtype == torch.float32 and tiles <= 64 and K >= 2048:
        splits = int(min(64, max(2, 256 // tiles), K // 512))
        key = "splitk/" + str(a.device)
        if key not in _ws or _ws[key].numel() < splits * M * N:
            _ws[key] = torch.empty(max(splits * M * N, 1 << 22), dtype=torch.float32, device=a.device)
        ws = _ws[key]
        _lib.check(_timed("linear_splitk_%dx%d" % (K, N), _lib.lib().mla_linear_splitk, _p(a), a.stride(0), _p(w), w.stride(0), _p(b),
                          _p(out), N, M, N, K, int(relu), splits, _p(ws), ws.numel(), _lib.stream_ptr()))
        return out
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


# ----------------------------------------------------------------- training-step kernels ----

class Dist:
    """Data-parallel context of the training step: SyncBN statistics, gradient buffer, loss and hit count are summed over
    the ranks of a ``torch.distributed`` process group.

    backend "nccl" (production: RCCL over xGMI): the sums run through the C ABI, ``mla_allreduce_flat`` on a communicator
    this object creates with ``mla_comm_init_rank`` (the 128-byte id travels from rank 0 through torch.distributed). The
    library binds the RCCL PyTorch has already loaded; MLA_DIST_COLLECTIVE=torch selects ``torch.distributed.all_reduce``
    on the same device buffers instead. backend "gloo" (CPU rehearsals, two ranks sharing one test GPU): host copy.
    world == 1: every all-reduce is a no-op unless ``always`` (or MLA_DIST_ALWAYS=1) asks for the collectives anyway --
    a one-rank sum leaves the data unchanged, so the production branch can be exercised on a one-GPU box."""

    def __init__(self, group=None, always=None, sync_bn=True):
        """sync_bn: train-mode BatchNorm statistics are summed over the ranks (the step then equals the reference's single-process
        step on the GLOBAL batch, train.py:119-142). False = per-shard statistics, the usual DistributedDataParallel semantics
        (SURVEY.md section 8e allows either): no statistics all-reduce, the result equals the reference run per shard with
        gradients averaged. trace: set to a list to record (tag, bytes, start event, end event) of every collective (bench.py)."""
        import os
        import torch.distributed as dist
        self.group, self.comm = group, None
        self.sync_bn, self.trace = sync_bn, None
        inited = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if inited else 1
        self.rank = dist.get_rank(group) if inited else 0
        self.backend = dist.get_backend(group) if inited else None
        always = (os.environ.get("MLA_DIST_ALWAYS") == "1") if always is None else always
        self.active = inited and (self.world > 1 or always)
        self.via = None
        self.fallback = None
        if self.active and self.backend == "nccl":
            self.via = os.environ.get("MLA_DIST_COLLECTIVE", "abi")
            if self.via == "abi":
                # every rank tries; the ranks then agree (one MIN all-reduce through torch.distributed) on whether ALL of them have a
                # communicator. If one could not make it (an RCCL without the entry points the library binds, a failed init), all of
                # them switch to torch.distributed.all_reduce on the same device buffers and say so (describe()["fallback"]): a
                # data-parallel run never ends up with ranks on different transports.
                err = None
                try:
                    self._init_comm(dist)
                except Exception as e:                      # noqa: BLE001 -- reported through describe(), the run continues on torch's transport
                    err = "%s: %s" % (type(e).__name__, e)
                ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device()))
                dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
                if int(ok.item()) == 0:
                    self.close()
                    self.via = "torch"
                    self.fallback = err or "another rank could not create its communicator through the C ABI"
        elif self.active:
            self.via = "host"

    def _init_comm(self, dist):
        dev = torch.device("cuda", torch.cuda.current_device())
        uid, err0 = torch.zeros(128, dtype=torch.uint8), None
        if self.rank == 0:
            buf = (ctypes.c_char * 128)()
            try:
                _lib.check(_lib.lib().mla_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p)))
                uid = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
            except Exception as e:                          # noqa: BLE001 -- the other ranks are waiting in the broadcast below: send them zeros
                err0 = e
        uid = uid.to(dev)
        src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
        dist.broadcast(uid, src=src, group=self.group)
        raw = bytes(uid.cpu().numpy().tobytes())
        if err0 is not None or not any(raw):
            raise err0 or RuntimeError("rank 0 could not obtain an RCCL unique id through the C ABI")
        comm = ctypes.c_void_p()
        _lib.check(_lib.lib().mla_comm_init_rank(ctypes.byref(comm), self.world, ctypes.c_char_p(raw), self.rank))
        self.comm = comm
        # destroyed by close() or at interpreter exit while torch and RCCL are still loaded -- never from __del__, which may
        # run during teardown after either is gone
        import atexit
        import weakref
        ref = weakref.ref(self)
        atexit.register(lambda: ref() is not None and ref().close())

    def close(self):
        if self.comm is not None:
            torch.cuda.synchronize()
            _lib.check(_lib.lib().mla_comm_destroy(self.comm))
            self.comm = None


    def _reduce(self, t):
        """The collective itself, whatever the world size."""
        import torch.distributed as dist
        if self.via == "abi":
            assert t.is_cuda and t.is_contiguous()
            code = {torch.float32: _lib.F32, torch.float64: _lib.F64, torch.int32: _lib.I32}[t.dtype]
            _lib.check(_lib.lib().mla_allreduce_flat(_p(t), t.numel(), code, self.comm, _lib.stream_ptr()))
        elif self.via == "torch":
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        elif t.is_cuda:                                   # gloo: through the host
            host = t.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(host)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def all_reduce_sum(self, t, tag="other"):
        """In-place sum over the group (see the class docstring for the transport)."""
        if not self.active:
            return t
        if self.trace is None:
            return self._reduce(t)
        e0, e1 = _event(), _event()
        e0.record()
        self._reduce(t)
        e1.record()
        self.trace.append((tag, t.numel() * t.element_size(), e0, e1))
        return t

    @property
    def bn_active(self):
        return self.active and self.sync_bn

    @property
    def bn_world(self):
        """Number of shards whose rows enter one BatchNorm statistic."""
        return self.world if self.bn_active else 1

    def ranks_reported(self):
        """The rank count as the TRANSPORT reports it (ncclCommCount of this object's communicator, or the process group's size):
        evidence that the collectives really span the ranks the launcher started."""
        if self.comm is not None:
            n = ctypes.c_int(-1)
            _lib.check(_lib.lib().mla_comm_count(self.comm, ctypes.byref(n)))
            return int(n.value)
        import torch.distributed as dist
        return dist.get_world_size(self.group) if self.active else 1

    def describe(self):
        origin = _lib.lib().mla_comm_library_origin().decode() if self.via == "abi" else None
        return {"active": bool(self.active), "backend": self.backend, "transport": {"abi": "mla_allreduce_flat (C ABI -> ncclAllReduce)",
                "torch": "torch.distributed.all_reduce", "host": "torch.distributed.all_reduce through host copies (gloo rehearsal)",
                None: None}[self.via], "rccl_library_origin": origin, "ranks": self.ranks_reported(), "sync_bn": bool(self.sync_bn),
                "fallback": getattr(self, "fallback", None)}


LOCAL = None          # set lazily (torch.distributed may not be initialised at import time)


def _local():
    global LOCAL
    if LOCAL is None:
        LOCAL = Dist.__new__(Dist)
        LOCAL.group, LOCAL.world, LOCAL.rank, LOCAL.active, LOCAL.comm, LOCAL.via = None, 1, 0, False, None, None
        LOCAL.sync_bn, LOCAL.trace, LOCAL.backend, LOCAL.fallback = True, None, None, None
    return LOCAL


def dropout_mask(n, seed, stream_id, offset, p_drop, device):
    """uint8 keep-mask from the counter-based generator of the library (mla_dropout_mask; numpy restatement: weights.keep_mask)."""
    out = torch.empty(n, dtype=torch.uint8, device=device)
    _lib.check(_lib.lib().mla_dropout_mask(_p(out), n, int(seed) & (2 ** 64 - 1), int(stream_id) & (2 ** 64 - 1), int(offset), float(p_drop),
                                           _lib.stream_ptr()))
    return out


def bn_stats_sync(x, mode, period, dist, running_mean=None, running_var=None, momentum=-1.0, also=(), tracked=None):
    """bn_stats with the (sum, sum of squares) all-reduced over the data-parallel group. `also`: further (running_mean,
    running_var, momentum) triples to update from the SAME statistics (two BatchNorms fed by one tensor, model.py:239-240:
    one pass over x and one all-reduce instead of two). tracked: the BatchNorm's num_batches_tracked (int64 device scalar), advanced
    by the finishing kernel; `also` entries are (running_mean, running_var, momentum, tracked)."""
    _chk(x, torch.float32)
    rows, cols = x.shape
    ch = period if mode == 0 else cols
    sums = torch.empty(2 * ch, dtype=torch.float64, device=x.device)
    L = _lib.lib()
    mean = torch.empty(ch, dtype=torch.float32, device=x.device)
    var = torch.empty(ch, dtype=torch.float32, device=x.device)
    count = (rows // period * cols if mode == 0 else rows) * dist.bn_world
    if dist.bn_active:
        _lib.check(L.mla_bn_stats_sums(_p(x), rows, cols, x.stride(0), mode, period, _p(_workspace(x.device)), _p(sums), _lib.stream_ptr()))
        dist.all_reduce_sum(sums, "syncbn_fwd")
        _lib.check(L.mla_bn_stats_finish(_p(sums), ch, float(count), _p(mean), _p(var), _p(running_mean), _p(running_var),
                                         float(momentum), _p(tracked), _lib.stream_ptr()))
    else:               # nothing to exchange between the two stages: one launch fewer (same bits)
        _lib.check(L.mla_bn_stats_fused(_p(x), rows, cols, x.stride(0), mode, period, _p(_workspace(x.device)), _p(sums), _p(mean), _p(var),
                                        _p(running_mean), _p(running_var), float(momentum), _p(tracked), _lib.stream_ptr()))
    for rm, rv, mom, trk in also:
        scratch_m, scratch_v = torch.empty_like(mean), torch.empty_like(var)
        _lib.check(L.mla_bn_stats_finish(_p(sums), ch, float(count), _p(scratch_m), _p(scratch_v), _p(rm), _p(rv), float(mom), _p(trk),
                                         _lib.stream_ptr()))
    return mean, var


def bn_backward(x, dy, yout, act, drop_scale, mode, period, mean, var, gamma, dist, dgamma, dbeta, want_dx=True,
                dx=None, accumulate=False, batch_stats=True):
    """BatchNorm backward through the fused activation/dropout. Writes dgamma/dbeta (local parts) and returns dx (global-batch
    exact under data parallelism). batch_stats=False: the forward normalised with FIXED statistics (eval mode: mean / var are the
    running buffers), so dx = gamma * rstd * g without the two batch-mean terms -- the same kernels with zero sums for dx."""
    _chk(x, torch.float32)
    rows, cols = x.shape
    ch = period if mode == 0 else cols
    L = _lib.lib()
    local = torch.empty(2 * ch, dtype=torch.float64, device=x.device)
    ld_y = yout.stride(0) if yout is not None else 0
    _lib.check(L.mla_bn_bwd_sums(_p(x), x.stride(0), _p(dy), dy.stride(0), _p(yout), ld_y, act, float(drop_scale), rows, cols, mode,
                                 period, _p(mean), _p(var), BN_EPS, _p(_workspace(x.device)), _p(local), _lib.stream_ptr()))
    glob = local
    if not batch_stats:
        glob = torch.zeros_like(local)
    elif dist.bn_active:
        glob = dist.all_reduce_sum(local.clone(), "syncbn_bwd")
    count = (rows // period * cols if mode == 0 else rows) * dist.bn_world
    if want_dx and dx is None:
        dx = torch.empty((rows, cols), dtype=torch.float32, device=x.device)
    _lib.check(L.mla_bn_bwd_apply(_p(x), x.stride(0), _p(dy), dy.stride(0), _p(yout), ld_y, act, float(drop_scale), rows, cols, mode,
                                  period, _p(mean), _p(var), _p(gamma), BN_EPS, _p(glob), _p(local), float(count),
                                  _p(dx) if want_dx else None, dx.stride(0) if want_dx else 0, int(accumulate), _p(dgamma),
                                  _p(dbeta), _lib.stream_ptr()))
    return dx


def attention_pool_bwd(dy, att, cla, bags, T, K):
    du_v, du_f = torch.empty_like(att), torch.empty_like(att)
    _lib.check(_lib.lib().mla_attention_pool_bwd(_p(dy), dy.stride(0), _p(att), _p(cla), bags, T, K, _p(du_v), _p(du_f), _lib.stream_ptr()))
    return du_v, du_f


def linear_small_bwd(a, w, dz, dw, db):
    M, K = a.shape
    N = w.shape[0]
    da = torch.empty((M, K), dtype=torch.float32, device=a.device)
    _lib.check(_lib.lib().mla_linear_small_bwd(_p(a), a.stride(0), _p(w), w.stride(0), _p(dz), dz.stride(0), M, N, K, _p(da), K,
                                               _p(dw), _p(db), _lib.stream_ptr()))
    return da


def transpose_padded(x):
    """(R, C) f32 | bf16 -> (C, R~) with zero padding (R~ = R rounded up to a 16-byte row), K-contiguous for the MFMA GEMM."""
    _chk(x)
    R, C = x.shape
    if x.dtype == torch.bfloat16:
        ld = (R + 7) // 8 * 8
        out = torch.empty((C, ld), dtype=torch.bfloat16, device=x.device)
        _lib.check(_lib.lib().mla_transpose_bf16(_p(x), x.stride(0), _p(out), ld, R, C, _lib.stream_ptr()))
        return out
    _chk(x, torch.float32)
    ld = (R + 3) // 4 * 4
    out = torch.zeros((C, ld), dtype=torch.float32, device=x.device) if ld != R else torch.empty((C, ld), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mla_transpose_f32(_p(x), x.stride(0), _p(out), ld, R, C, _lib.stream_ptr()))
    return out


def col_sum(x, out):
    _chk(x)
    rows, cols = x.shape
    key = "colsum/" + str(x.device)
    if key not in _ws or _ws[key].numel() < 64 * cols:
        _ws[key] = torch.empty(64 * max(cols, 4096), dtype=torch.float64, device=x.device)
    ws = _ws[key]
    fn = _lib.lib().mla_col_sum_bf16 if x.dtype == torch.bfloat16 else _lib.lib().mla_col_sum
    _lib.check(fn(_p(x), x.stride(0), rows, cols, _p(ws), _p(out), _lib.stream_ptr()))
    return out


def axpy(a, x, y):
    _lib.check(_lib.lib().mla_axpy(float(a), _p(x), _p(y), x.numel(), _lib.stream_ptr()))
    return y


def check_labels(labels, K):
    """nn.CrossEntropyLoss raises on a target outside [0, K) (train.py:372 keeps the default ignore_index=-100, a value the
    reference's dataset never produces -- labels are class indices, dataset.py:60-75 -- and which is refused here like any other
    negative label rather than silently dropped from the mean). Host tensors are checked
    here before the upload (no device sync); for device tensors the kernel refuses to index with such a label and reports
    it through its second counter (`hits[1]` = number of such labels) + a NaN loss, which `raise_on_bad_labels` turns into the
    same exception. The two counters are separate so that their sum over data-parallel ranks cannot cancel."""
    if not labels.is_cuda and labels.numel():
        lo, hi = int(labels.min()), int(labels.max())
        if lo < 0 or hi >= K:
            raise IndexError("Target %d is out of bounds." % (lo if lo < 0 else hi))


def raise_on_bad_labels(hits):
    """Call where the step's result is read on the host anyway (train.py:141-142): `hits` is cross_entropy's pair
    [n_correct, n_bad_labels] (summed over the ranks by a data-parallel step). Returns n_correct."""
    n, bad = (int(v) for v in hits.tolist())
    if bad > 0:
        raise IndexError("Target out of bounds: %d label(s) outside [0, num_classes)." % bad)
    return n


def cross_entropy(scores, labels, inv_total, want_grad=True):
    """CrossEntropyLoss(mean over the GLOBAL batch) on (B, K) scores: (loss, dscores, hits) with hits = int32 [n_correct,
    n_bad]; n_bad > 0 (and a NaN loss): that many labels were outside [0, K) -- see check_labels."""
    _chk(scores, torch.float32); _chk(labels, torch.int64)
    B, K = scores.shape
    loss = torch.empty(1, dtype=torch.float32, device=scores.device)
    hits = torch.empty(2, dtype=torch.int32, device=scores.device)
    d = torch.empty_like(scores) if want_grad else None
    _lib.check(_lib.lib().mla_cross_entropy(_p(scores), scores.stride(0), _p(labels), B, K, float(inv_total), _p(loss), _p(d),
                                            K, _p(hits), _lib.stream_ptr()))
    return loss, d, hits


def dropout_mask_dev(n, seed, stream_base, counter, offset, p_drop, device, out=None):
    """dropout_mask with stream_id = stream_base + counter[0] + 1 read on the device (graph-captured training step)."""
    assert counter.dtype == torch.int64 and counter.is_cuda
    if out is None:
        out = torch.empty(n, dtype=torch.uint8, device=device)
    _lib.check(_lib.lib().mla_dropout_mask_dev(_p(out), n, int(seed) & (2 ** 64 - 1), int(stream_base) & (2 ** 64 - 1), _p(counter), int(offset),
                                               float(p_drop), _lib.stream_ptr()))
    return out


def adam_prepare(scal, lr, beta1, beta2, step):
    """Write Adam's step-dependent scalars for step `step` into the device pair adam_step_dev reads (an ordinary launch, enqueued in
    front of a graph replay)."""
    assert scal.dtype == torch.float32 and scal.is_cuda and scal.numel() == 2
    _lib.check(_lib.lib().mla_adam_prepare(_p(scal), float(lr), float(beta1), float(beta2), int(step), _lib.stream_ptr()))


def adam_step_dev(p, g, m, v, beta1, beta2, eps, scal, counter):
    """adam_step with its step-dependent scalars read from `scal` (adam_prepare), then counter += 1 (both graph-capturable)."""
    L = _lib.lib()
    _lib.check(L.mla_adam_step_dev(_p(p), _p(g), _p(m), _p(v), p.numel(), float(beta1), float(beta2), float(eps), _p(scal), _lib.stream_ptr()))
    _lib.check(L.mla_counter_add(_p(counter), 1, _lib.stream_ptr()))


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step):
    _lib.check(_lib.lib().mla_adam_step(_p(p), _p(g), _p(m), _p(v), p.numel(), float(lr), float(beta1), float(beta2), float(eps),
                                        int(step), _lib.stream_ptr()))


# ------------------------------------------------------------------ finetune (CNN gradients) ----

def conv3x3(x, w_packed, b, cout, pool, act):
    """Generic conv entry (training forward without fused pool, and dgrad), f32 or bf16. x NHWC."""
    _chk(x); _chk(w_packed, x.dtype)
    n, H, W_, cin = x.shape
    out = torch.empty((n, H // 2, W_ // 2, cout) if pool else (n, H, W_, cout), dtype=x.dtype, device=x.device)
    _lib.check(_timed("conv3x3_%d_%d" % (cin, cout), _lib.lib().mla_conv3x3, _p(x), _p(w_packed), _p(b), _p(out), n, H, W_, cin, cout,
                      int(pool), int(act), DT[x.dtype], _lib.stream_ptr()))
    return out


def conv3x3_train(x, w_packed, b, cout):
    """Training forward of a pooled layer: (pre-pool post-ReLU activation, its 2x2 max-pool) from one kernel."""
    _chk(x); _chk(w_packed, x.dtype)
    n, H, W_, cin = x.shape
    a = torch.empty((n, H, W_, cout), dtype=x.dtype, device=x.device)
    pooled = torch.empty((n, H // 2, W_ // 2, cout), dtype=x.dtype, device=x.device)
    _lib.check(_timed("conv3x3_%d_%d" % (cin, cout), _lib.lib().mla_conv3x3_train, _p(x), _p(w_packed), _p(b), _p(a), _p(pooled), n, H, W_,
                      cin, cout, DT[x.dtype], _lib.stream_ptr()))
    return a, pooled


def conv3x3_train_codes(x, w_packed, b, cout):
    """Training forward of a pooled layer, compact form: (window codes uint8 (n, H/2, W/2, cout), pooled activation). A code is the
    position 0..3 of the first maximum of the 2x2 window (MaxPool2d's order), or 4 where the ReLU is off -- one byte instead of the
    four pre-pool activations (8 bytes in bf16) that conv3x3_train keeps for the backward."""
    _chk(x); _chk(w_packed, x.dtype)
    n, H, W_, cin = x.shape
    codes = torch.empty((n, H // 2, W_ // 2, cout), dtype=torch.uint8, device=x.device)
    pooled = torch.empty((n, H // 2, W_ // 2, cout), dtype=x.dtype, device=x.device)
    _lib.check(_timed("conv3x3_%d_%d" % (cin, cout), _lib.lib().mla_conv3x3_train_codes, _p(x), _p(w_packed), _p(b), _p(codes), _p(pooled),
                      n, H, W_, cin, cout, DT[x.dtype], _lib.stream_ptr()))
    return codes, pooled


def pool_bwd_codes(codes, d_out, db=None):
    """dZ (n, H, W, C) bf16 at the pre-pool resolution from the window codes and the pooled gradient; db: bias gradient on the way."""
    assert codes.dtype == torch.uint8 and codes.is_cuda and codes.is_contiguous()
    _chk(d_out, torch.bfloat16)
    n, HO, WO, C = codes.shape
    assert tuple(d_out.shape) == tuple(codes.shape)
    dz = torch.empty((n, 2 * HO, 2 * WO, C), dtype=torch.bfloat16, device=codes.device)
    key = "biasws16/" + str(codes.device)
    if key not in _ws:
        _ws[key] = torch.empty(int(_lib.lib().mla_relu_pool_bwd_bf16_workspace_bytes()) // 8, dtype=torch.float64, device=codes.device)
    _lib.check(_timed("relu_pool_bwd", _lib.lib().mla_pool_bwd_codes_bf16, _p(codes), _p(d_out), _p(dz), n, 2 * HO, 2 * WO, C, _p(_ws[key]),
                      _p(db), _lib.stream_ptr()))
    return dz


def repack_dgrad(w, dtype=torch.float32):
    """(Cout, Cin, 3, 3) f32 -> (Cin, 9, Cout) flipped, in `dtype`: weights of the transposed convolution."""
    _chk(w, torch.float32)
    cout, cin = w.shape[0], w.shape[1]
    out = torch.empty((cin, 9, cout), dtype=dtype, device=w.device)
    fn = _lib.lib().mla_conv_repack_dgrad_bf16 if dtype == torch.bfloat16 else _lib.lib().mla_conv_repack_dgrad
    _lib.check(fn(_p(w), cout, cin, _p(out), _lib.stream_ptr()))
    return out


def maxpool2x2(a):
    _chk(a)
    n, H, W_, C = a.shape
    out = torch.empty((n, H // 2, W_ // 2, C), dtype=a.dtype, device=a.device)
    fn = _lib.lib().mla_maxpool2x2_bf16 if a.dtype == torch.bfloat16 else _lib.lib().mla_maxpool2x2
    _lib.check(_timed("maxpool", fn, _p(a), _p(out), n, H, W_, C, _lib.stream_ptr()))
    return out


def relu_pool_bwd(a, d_out, pool, db=None, bf16=False):
    """dZ at a's resolution from the gradient of relu(.) [pool == 0] or maxpool(relu(.)) [pool == 1]; db (C,): also the
    bias gradient = column sums of dZ, accumulated in the same pass. bf16 (or a bf16 `a`): dZ in bf16 (a / d_out both bf16,
    or both f32 for the last Linear whose output is f32)."""
    _chk(a); _chk(d_out, a.dtype)
    a4 = a if a.dim() == 4 else a.reshape(a.shape[0], 1, 1, -1)
    n, H, W_, C = a4.shape
    if bf16 or a.dtype == torch.bfloat16:
        dz = torch.empty(a.shape, dtype=torch.bfloat16, device=a.device)
        key = "biasws16/" + str(a.device)
        if key not in _ws:
            _ws[key] = torch.empty(int(_lib.lib().mla_relu_pool_bwd_bf16_workspace_bytes()) // 8, dtype=torch.float64, device=a.device)
        _lib.check(_timed("relu_pool_bwd", _lib.lib().mla_relu_pool_bwd_bf16, _p(a), DT[a.dtype], _p(d_out), DT[d_out.dtype], _p(dz), n, H, W_, C,
                          int(pool), _p(_ws[key]), _p(db), _lib.stream_ptr()))
        return dz
    _chk(a, torch.float32)
    dz = torch.empty_like(a)
    if db is None:
        _lib.check(_lib.lib().mla_relu_pool_bwd(_p(a), _p(d_out), _p(dz), n, H, W_, C, int(pool), _lib.stream_ptr()))
        return dz
    _chk(db, torch.float32)
    key = "biasws/" + str(a.device)
    if key not in _ws:
        _ws[key] = torch.empty(int(_lib.lib().mla_relu_pool_bwd_bias_workspace_bytes()) // 8, dtype=torch.float64, device=a.device)
    _lib.check(_timed("relu_pool_bwd", _lib.lib().mla_relu_pool_bwd_bias, _p(a), _p(d_out), _p(dz), n, H, W_, C, int(pool), _p(_ws[key]), _p(db),
                      _lib.stream_ptr()))
    return dz


_wg_ws = {}


def conv_wgrad(dz, a_in, dw):
    _chk(dz); _chk(a_in, dz.dtype)
    n, H, W_, cout = dz.shape
    cin = a_in.shape[3]
    key = str(dz.device)
    if key not in _wg_ws:
        _wg_ws[key] = torch.empty(int(_lib.lib().mla_conv_wgrad_workspace_floats()), dtype=torch.float32, device=dz.device)
    ws = _wg_ws[key]
    assert dw.is_contiguous() and tuple(dw.shape) == (cout, cin, 3, 3) and dw.dtype == torch.float32
    fn = _lib.lib().mla_conv_wgrad_bf16 if dz.dtype == torch.bfloat16 else _lib.lib().mla_conv_wgrad
    _lib.check(_timed("wgrad_%d_%d" % (cin, cout), fn, _p(dz), _p(a_in), n, H, W_, cin, cout, _p(ws), ws.numel(), _p(dw), _lib.stream_ptr()))


def conv1_bwd(x, w, b, d_pooled, dw, db):
    _chk(x, torch.float32); _chk(d_pooled)
    n = x.shape[0]
    ws = torch.empty(int(_lib.lib().mla_conv1_bwd_workspace_floats()), dtype=torch.float32, device=x.device)
    fn = _lib.lib().mla_conv1_bwd_bf16 if d_pooled.dtype == torch.bfloat16 else _lib.lib().mla_conv1_bwd
    _lib.check(_timed("conv1_bwd", fn, _p(x), _p(w), _p(b), _p(d_pooled), n, _p(ws), _p(dw), _p(db), _lib.stream_ptr()))
