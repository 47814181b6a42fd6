"""Drop-in for the reference's ``model.py`` (vggish branch): ``Ensemble``, ``Input``, ``CNN``,
``CnnFlatten``, ``EmbeddedMapping``, ``AttentionModule``, ``MultiLevelAttention`` and
``set_requires_grad`` with the same constructor arguments, attribute names and ``state_dict``
keys (model.py:14, :68, :107, :179, :202, :228, :248, :272), executed by HIP kernels.

Reproduced on purpose (SURVEY.md section 7 "quirks"): ``fcv`` feeds both attention branches
and ``fcf`` is a dead parameter (model.py:230-238); BatchNorm1d(T) treats the time slot as
channel; the outputs are sigmoids that train.py feeds to CrossEntropyLoss; ``Input`` is a
reshape, never a transpose (model.py:98-99). The ``resnet`` branches are outside the hot path
 and raise.
"""

from typing import Dict, List, Union

import torch
from torch import nn

from . import differentiable, mla_train, ops
from .params import *  # noqa: F401,F403  (T, H, K, DR, M_VGGISH, M_VGGISH_JB, S_VGGISH_SHAPE: model.py:9)
from .torchvggish.vggish import Linear, VGGish


class BatchNorm1d(nn.Module):
    """Parameter / buffer holder with torch.nn.BatchNorm1d's names (weight, bias, running_mean,
    running_var, num_batches_tracked), defaults eps 1e-5, momentum 0.1."""

    def __init__(self, num_features):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, 1e-5, 0.1
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def forward(self, x):
        raise RuntimeError("BatchNorm1d is executed by the fused HIP head of its parent module")


class Dropout(nn.Module):
    """nn.Dropout(p) stand-in. ``mask`` (uint8 keep-mask, 1 = keep) may be injected for reproducible runs; otherwise a
    fresh mask is drawn per training forward by the library's counter-based generator (``mla_dropout_mask``): element i
    of the GLOBAL batch is kept iff hash24(seed, stream, i) >= p * 2^24, with seed = ``torch.initial_seed()`` at
    construction (so ``torch.manual_seed`` makes runs repeatable, as with nn.Dropout) and stream = (this module's
    ordinal, call count). A data-parallel rank passes the offset of its shard: N ranks draw the masks of one process."""

    _instances = 0

    def __init__(self, p=0.5):
        super().__init__()
        self.p, self.mask = p, None
        self.seed, self.calls = torch.initial_seed(), 0
        Dropout._instances += 1
        self.ordinal = Dropout._instances

    def keep_mask(self, numel, device, offset=0):
        if self.mask is not None:
            m = self.mask.to(device=device, dtype=torch.uint8).reshape(-1).contiguous()
            assert m.numel() == numel
            return m
        self.calls += 1
        return ops.dropout_mask(numel, self.seed, (self.ordinal << 32) + self.calls, offset, self.p, device)

    def keep_mask_dev(self, numel, device, offset, counter, base):
        """The mask of call number base + counter[0] + 1 of this module, the call number read ON THE DEVICE: what a HIP graph of
        the training step records, so that every replay draws the next mask of the sequence keep_mask() would have drawn."""
        return ops.dropout_mask_dev(numel, self.seed, (self.ordinal << 32) + base, counter, offset, self.p, device)

    def forward(self, x):
        raise RuntimeError("Dropout is fused into the HIP BatchNorm/ReLU kernel of its parent module")


class Ensemble(nn.Module):
    def __init__(self, input_conf: str, cnn_conf: Dict[str, Union[str, int]], model_conf: List[int], device,
                 precision: str = "f32"):
        super().__init__()
        self.cnn_type = cnn_conf["cnn_type"]
        self.just_bottlenecks = cnn_conf["just_bottlenecks"]
        self.num_classes = cnn_conf["num_classes"]
        if self.cnn_type == "vggish" and self.just_bottlenecks:
            self.emb_input_size = M_VGGISH_JB
        elif self.cnn_type == "vggish" and not self.just_bottlenecks:
            self.emb_input_size = M_VGGISH
        elif self.cnn_type == "resnet":
            raise Exception("cnn_type 'resnet' is outside the MI355X hot path (torchvision ResNet-50); use 'vggish'.")
        else:
            raise Exception("CNN type is not valid.")
        self.input = Input(input_conf=input_conf, cnn_type=self.cnn_type, device=device)
        self.mla = MultiLevelAttention(model_conf, self.emb_input_size)
        self.cnn = CNN(**cnn_conf, precision=precision)

    def set_precision(self, precision):
        self.cnn.set_precision(precision)
        return self

    def forward(self, x):
        x_proc = self.input(x)
        features = self.cnn(x_proc)
        out = self.mla(features.reshape(-1, T, self.emb_input_size))
        return out

    def forward_waveforms(self, pcm):
        """Fused online path of the north star: (B, n_samples) 16 kHz PCM on the device (float32
        or int16), one bag per row with exactly T examples -> (B, K) scores. The front-end writes
        the examples directly in the CNN's compute dtype."""
        from . import frontend
        dtype = torch.bfloat16 if self.cnn.precision == "bf16" else torch.float32
        ex = frontend.waveforms_to_examples(pcm, out_dtype=dtype)
        assert ex.shape[0] == pcm.shape[0] * T, "each waveform must yield exactly T examples"
        features = self.cnn(ex)
        return self.mla(features.reshape(-1, T, self.emb_input_size))

    def stream_waveforms(self, host_batches):
        """Host-resident PCM: iterate over (B, n_samples) float32 / int16 tensors in PINNED host memory and yield the
        (B, K) scores of each. The copy of batch i+1 runs on its own HIP stream while batch i computes (two device
        buffers), so a steady stream is bound by max(copy, compute), not their sum. Scores are yielded after the batch's
        compute has been enqueued; they are ordinary tensors on the current stream."""
        dev = next(self.parameters()).device
        cur = torch.cuda.current_stream(dev)
        copy = torch.cuda.Stream(device=dev)
        it = iter(host_batches)

        def upload(host):
            assert host.is_pinned(), "stream_waveforms overlaps asynchronous copies: the host tensors must be pinned"
            with torch.cuda.stream(copy):
                d = host.to(dev, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(copy)
            return d, ev

        nxt = next(it, None)
        pending = upload(nxt) if nxt is not None else None
        while pending is not None:
            d, ev = pending
            nxt = next(it, None)
            pending = upload(nxt) if nxt is not None else None      # in flight while `d` computes
            cur.wait_event(ev)
            out = self.forward_waveforms(d)
            d.record_stream(cur)                                     # the caching allocator must not recycle `d` early
            yield out

    def capture_waveforms(self, pcm):
        """Capture forward_waveforms for inputs shaped like `pcm` into a HIP graph (eval mode only) and return a
        callable that replays it: ~45 kernel launches per step become one graph launch, which is what small
        batches (about 1 000 clips, where a step is ~1.5 ms of GPU work) need."""
        return GraphedWaveforms(self, pcm)


class GraphedWaveforms:
    """forward_waveforms of an eval-mode Ensemble captured once into a HIP graph. The C-ABI kernels are plain
    launches on torch's current stream, so torch's capture records them; scratch tensors allocated while
    capturing live in the graph's private pool. Call with a tensor of the captured shape/dtype."""

    def __init__(self, model, pcm):
        assert not model.training, "graph capture covers the eval-mode forward (train-mode statistics sync with the host)"
        self.model = model
        self.static_in = pcm.clone()
        side = torch.cuda.Stream(device=pcm.device)
        side.wait_stream(torch.cuda.current_stream(pcm.device))
        with torch.cuda.stream(side), torch.no_grad():       # weight repacks / caches / workspaces are built here
            for _ in range(2):
                model.forward_waveforms(self.static_in)
        torch.cuda.current_stream(pcm.device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode="relaxed"), torch.no_grad():
            self.static_out = model.forward_waveforms(self.static_in)

    def __call__(self, pcm):
        if pcm.data_ptr() != self.static_in.data_ptr():
            assert pcm.shape == self.static_in.shape and pcm.dtype == self.static_in.dtype
            self.static_in.copy_(pcm, non_blocking=True)
        self.graph.replay()
        return self.static_out


class Input(nn.Module):
    def __init__(self, input_conf, cnn_type, device):
        super().__init__()
        self.conf, self.device, self.cnn_type = input_conf, device, cnn_type

    def forward(self, x):
        if self.cnn_type == "vggish":
            return x.reshape((-1, 1, S_VGGISH_SHAPE[0], S_VGGISH_SHAPE[1]))
        elif self.cnn_type == "resnet":
            raise Exception("cnn_type 'resnet' is outside the MI355X hot path.")
        else:
            raise Exception("CNN type is not valid.")


class CNN(nn.Module):
    def __init__(self, cnn_type="vggish", num_classes=10, use_pretrained=True, just_bottlenecks=False,
                 cnn_trainable=False, first_cnn_layer_trainable=False, in_channels=3, precision="f32"):
        super().__init__()
        self.precision = precision
        if cnn_type == "vggish":
            model_urls = {"vggish": "https://github.com/harritaylor/torchvggish/releases/download/v0.1/vggish-10086976.pth"}
            self.cnn_model = VGGish(urls=model_urls, pretrained=use_pretrained, preprocess=False, postprocess=False,
                                    progress=True, precision=precision)
            if not cnn_trainable:
                set_requires_grad(self.cnn_model, False)
            if just_bottlenecks:
                self.cnn_model = nn.Sequential(list(self.cnn_model.children())[0], CnnFlatten(cnn_type))
        elif cnn_type == "resnet":
            raise Exception("cnn_type 'resnet' is outside the MI355X hot path (torchvision ResNet-50); use 'vggish'.")
        else:
            raise Exception("Invalid CNN model name specified.")

    def set_precision(self, precision):
        self.precision = precision
        for m in self.cnn_model.modules():
            if hasattr(m, "precision"):
                m.precision = precision
        return self

    def forward(self, x):
        x = self.cnn_model(x)
        if x.dtype == torch.bfloat16:      # bf16 bottlenecks (just_bottlenecks=True) feed the f32 head
            if x.requires_grad:
                return differentiable.CastFn.apply(x)
            x = ops.merge_split(x.contiguous(), 512) if self.precision == "bf16x3" else ops.to_f32(x.contiguous())
        return x


class CnnFlatten(nn.Module):
    def __init__(self, cnn_type):
        super().__init__()
        self.cnn_type = cnn_type

    def forward(self, x):
        if self.cnn_type == "vggish":
            x = torch.transpose(x, 1, 3)
            x = torch.transpose(x, 1, 2)
            x = x.contiguous()           # no copy: VGGFeatures returns an NCHW view of NHWC memory
            x = x.view(x.size(0), -1)
        else:
            raise Exception("Invalid CNN model name specified.")
        return x


class EmbeddedMapping(nn.Module):
    def __init__(self, n_fc, is_first, emb_input_size):
        super().__init__()
        self.n_fc = n_fc
        self.norm0 = BatchNorm1d(T)
        if is_first:
            self.fc = nn.ModuleList([Linear(emb_input_size, H)] + [Linear(H, H) for _ in range(n_fc - 1)])
        else:
            self.fc = nn.ModuleList([Linear(H, H) for _ in range(n_fc)])
        self.dropouts = nn.ModuleList([Dropout(p=DR) for _ in range(n_fc)])
        self.norms = nn.ModuleList([BatchNorm1d(T) for _ in range(n_fc)])

    def forward(self, x):
        return mla_train.embedded_mapping_forward(self, x, None)


class AttentionModule(nn.Module):
    def __init__(self):
        super().__init__()
        self.fcv = Linear(H, K)
        self.fcf = Linear(H, K)          # never used by forward (model.py:237-238); kept for the state_dict
        self.normv = BatchNorm1d(T)
        self.normf = BatchNorm1d(T)

    def forward(self, h):
        y = torch.empty((h.shape[0], K), dtype=torch.float32, device=h.device)
        mla_train.attention_forward(self, h, y, None)
        return y


class MultiLevelAttention(nn.Module):
    def __init__(self, model_conf, emb_input_size):
        super().__init__()
        self.model = model_conf
        self.embedded_mappings = nn.ModuleList(
            [EmbeddedMapping(model_conf[0], is_first=True, emb_input_size=emb_input_size)] +
            [EmbeddedMapping(n_layers, is_first=False, emb_input_size=emb_input_size) for n_layers in model_conf[1:]])
        self.attention_modules = nn.ModuleList([AttentionModule() for _ in model_conf])
        self.fc = Linear(len(model_conf) * K, K)
        self.norm = BatchNorm1d(K)

    def forward(self, x):
        if differentiable.wants_grad(self, x):
            # train.py:124-138 on the drop-in: outputs = clf(inputs); loss.backward() -- the head's HIP backward runs behind autograd,
            # in train mode (batch statistics, dropout) and in eval mode (running statistics) alike
            return differentiable.HeadFn.apply(self, x.float(), *self.parameters())
        return mla_train.mla_apply(self, x)


def set_requires_grad(model, value):
    for param in model.parameters():
        param.requires_grad = value
