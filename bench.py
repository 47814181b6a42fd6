#!/usr/bin/env python3
"""Headline benchmark: 0.96 s clips/s, wave -> class scores (VGGish + multi-level attention).

    python bench.py --gpus N --steps K --warmup W          (N > 1 from a plain `python`: this process spawns the N ranks
                                                            itself, one per GPU, BEFORE it touches the GPU; under
                                                            torch.distributed.run it is one of the ranks)
    python bench.py --mode train [--finetune] ...          (BASELINE configs 4/5: the data-parallel train.py step)

One step = one pass of the whole hot path over one batch of synthetic 16 kHz PCM that is
already resident in HBM: fused log-mel front-end -> conv stack -> FC embeddings -> MLA head,
BASELINE.json config 3: 1024 bags x 10 s per GPU (= 10 240 x 0.96 s clips), bf16 conv/FC on the
matrix cores, f32 front-end and head. Bags are independent in eval mode, so N GPUs run N
shards with no data-path collective (weak scaling); timing is max over ranks.

The JSON line also carries `roofline` (dominant kernel, measured live with events on the
launch stream) and, on rank 0 at N = 1, `cpu_baseline` (the oracle = CPU restatement of the
reference, timed on this host's cores on a bounded sample).
"""

import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "audio-classification-using-a-deep-cnn-combined-with-multi-level-attention_amd"

import numpy as np
import torch

T_BAG, N_SAMPLES = 10, 160000
CONV_MFLOP = {"conv1": 7.08, "conv2": 226.49, "conv3": 226.49, "conv4": 452.98, "conv5": 226.49, "conv6": 452.98}
FC_MFLOP = {"linear_12288x4096": 100.66, "linear_4096x4096": 33.55, "linear_4096x128": 1.05}
FE_BYTES = {torch.float32: 15360 * 4 + 96 * 64 * 4, torch.bfloat16: 15360 * 4 + 96 * 64 * 2}
CONV_DESC = {"conv2": "64->128 @48x32 +pool", "conv3": "128->256 @24x16", "conv4": "256->256 @24x16 +pool",
             "conv5": "256->512 @12x8", "conv6": "512->512 @12x8 +pool"}
# MI355X_MICROARCH.md: dense bf16 MFMA / f32 MFMA. bf16x3 is priced in ALGORITHMIC flops against the bf16 peak (it issues 3x)
PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3, "bf16x3": 2500.0}
PEAK_HBM_GBPS = 8000.0

# algorithmic HBM bytes per clip of a conv layer: input + output activations once (elem = bytes per element); weights are
# read once per launch and are negligible
CONV_BYTES = {"conv2": lambda e: (48 * 32 * 64 + 24 * 16 * 128) * e, "conv3": lambda e: (24 * 16 * 128 + 24 * 16 * 256) * e,
              "conv4": lambda e: (24 * 16 * 256 + 12 * 8 * 256) * e, "conv5": lambda e: (12 * 8 * 256 + 12 * 8 * 512) * e,
              "conv6": lambda e: (12 * 8 * 512 + 6 * 4 * 512) * e}


def kernel_averages(prof):
    per = {}
    for name, e0, e1 in prof:
        per.setdefault(name, []).append(e0.elapsed_time(e1) * 1e-3)
    return {k: sum(v) / len(v) for k, v in per.items()}


CNN_CONF = dict(cnn_type="vggish", num_classes=10, use_pretrained=False, just_bottlenecks=False,
                cnn_trainable=False, first_cnn_layer_trainable=False, in_channels=1)


def build_model(precision, device):
    W = importlib.import_module(PKG + ".weights")
    M = importlib.import_module(PKG + ".model")
    sd = W.make_state_dict(6, W.ensemble_shapes((2, 1), False))
    ens = M.Ensemble("repeat", CNN_CONF, [2, 1], device, precision=precision)
    ens.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    return ens.to(device).eval(), sd


def synth_pcm(bags, rank, device):
    """`bags` waveforms of 10 s: 16 distinct portable-seeded waveforms, tiled (values do not
    affect kernel time; distinct data keeps DVFS honest vs zeros)."""
    W = importlib.import_module(PKG + ".weights")
    base = torch.from_numpy(W.waveform(3000 + rank, N_SAMPLES, min(bags, 16)))
    reps = (bags + base.shape[0] - 1) // base.shape[0]
    return base.repeat(reps, 1)[:bags].contiguous().to(device)


def cpu_baseline(ens_sd, gpu_model, device, budget_s=15.0):
    """The oracle (kind 'port') on config C1 = 8 x 10 s -> 80 clips, repeated for ~budget_s."""
    from oracle import frontend as ofe
    from oracle import model as omodel
    W = importlib.import_module(PKG + ".weights")
    # the one-GPU box's CPU share is 16 cores; more torch threads than that only oversubscribe
    threads = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(threads)
    wav = W.waveform(1000, N_SAMPLES, 8)
    sd = omodel.to_torch(ens_sd)

    def one():
        ex = torch.as_tensor(ofe.batch_examples(wav.astype(np.float64))).float()
        with torch.no_grad():
            return omodel.ensemble_forward(sd, ex.reshape(8, T_BAG, 1, 96, 64))

    ref = one()
    t0, reps = time.perf_counter(), 0
    while reps < 3 or time.perf_counter() - t0 < budget_s:
        one()
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    out = {"value": 80.0 / dt, "unit": "clips/s", "cores": threads, "host_cpus": os.cpu_count(), "kind": "port",
           "sample": "config C1: 8 x 10 s waveforms -> 80 clips, oracle (numpy f64 front-end + torch-CPU f32 model), "
                     "%d repetitions, mean %.3f s each" % (reps, dt)}
    pcm = torch.from_numpy(wav).to(device)
    refn = ref.numpy()
    keep = gpu_model.cnn.precision
    for prec in ("f32", "bf16x3", "bf16"):
        got = gpu_model.set_precision(prec).forward_waveforms(pcm).cpu().numpy()
        out["parity_max_rel_%s" % prec] = float(np.abs(got - refn).max() / np.abs(refn).max())
    gpu_model.set_precision(keep)
    return out


def small_batch_leg(ens, rank, device, bags=102, steps=50):
    """BASELINE config 3 read literally ("batch 1024 clips"): 102 bags = 1 020 clips per step, ~1.45 ms of GPU work behind ~45
    launches, timed eagerly. (Rounds 1-2 also timed one HIP-graph replay per step: it measured 1-3 % SLOWER than eager at this size
    -- the step is bound by kernel time, not by launches -- and was dropped from the line; Ensemble.capture_waveforms stays for
    batches small enough to be launch-bound and is checked here for bit-identity with the eager result.)"""
    pcm = synth_pcm(bags, rank, device)
    out = {"clips_per_step": bags * T_BAG}
    with torch.no_grad():
        g = ens.capture_waveforms(pcm)
        ref = ens.forward_waveforms(pcm).clone()
        assert torch.equal(g(pcm), ref), "graph replay must reproduce the eager result"
        for _ in range(5):
            ens.forward_waveforms(pcm)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            ens.forward_waveforms(pcm)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        out["eager"] = {"ms_per_step": dt * 1e3, "clips_per_s": bags * T_BAG / dt}
    return out


def parity_mode_leg(ens, pcm, clips_per_step, ops, steps=5):
    """The same batch in the bf16x3 mode: still bf16 MFMA arithmetic, but every value carried as hi + lo bf16 planes and
    every product as three bf16 terms with f32 accumulation, which meets the north star's 1e-4 relative tolerance on the
    scores (the plain bf16 headline does not: `cpu_baseline.parity_max_rel_*`). Also the exact-f32 MFMA mode beside it.
    conv_stack_frac_algorithmic prices the layer's algorithmic flops against the mode's MFMA peak (bf16 peak for bf16x3);
    conv_stack_frac_issued counts the three bf16 products bf16x3 actually issues per term."""
    out = {}
    keep = ens.cnn.precision
    with torch.no_grad():
        for prec, n in (("bf16x3", steps), ("f32", 2)):
            ens.set_precision(prec)
            ens.forward_waveforms(pcm)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                ens.forward_waveforms(pcm)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            ops.profile = []
            ens.forward_waveforms(pcm)
            torch.cuda.synchronize()
            avg, ops.profile = kernel_averages(ops.profile), None
            conv_t = sum(avg[k] for k in avg if k in CONV_MFLOP)
            frac = clips_per_step * sum(CONV_MFLOP.values()) * 1e6 / conv_t / 1e12 / PEAK_TFLOPS[prec]
            out[prec] = {"ms_per_step": dt * 1e3, "clips_per_s": clips_per_step / dt, "conv_stack_ms": conv_t * 1e3,
                         "conv_stack_frac_algorithmic": frac, "conv_stack_frac_issued": frac * (3 if prec == "bf16x3" else 1)}
    ens.set_precision(keep)
    return out


def h2d_leg(pcm, step_s, clips_per_step, reps=5, ens=None):
    """The Python boundary also accepts host arrays (vggish_input.waveform_to_examples(np.ndarray)): the PCIe-inclusive
    rate for this batch from pinned host memory: serial (copy, then compute), the bound when the copy of batch i+1 overlaps
    the compute of batch i, and that overlap measured (Ensemble.stream_waveforms). Reported beside `value`, never as `value`."""
    host = torch.empty(pcm.shape, dtype=pcm.dtype).pin_memory()
    host.copy_(pcm)
    dst = torch.empty_like(pcm)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        dst.copy_(host, non_blocking=True)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    copy_s = sorted(ts)[len(ts) // 2]
    out = {"bytes": pcm.numel() * pcm.element_size(), "copy_ms": copy_s * 1e3, "GBps": pcm.numel() * pcm.element_size() / copy_s / 1e9,
           "clips_per_s_serial": clips_per_step / (copy_s + step_s), "clips_per_s_overlapped_bound": clips_per_step / max(copy_s, step_s),
           "pcm": str(pcm.dtype).replace("torch.", "") + ", pinned host memory"}
    if ens is not None:            # measured: Ensemble.stream_waveforms (copy stream + two device buffers) over 24 host batches
        n = 24                     # (the first batch's copy is not overlapped: 1 / n of the stream is pipeline fill)
        with torch.no_grad():
            for _ in ens.stream_waveforms([host] * 2):
                pass
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for o in ens.stream_waveforms([host] * n):
                pass
            torch.cuda.synchronize()
        out["clips_per_s_streamed"] = n * clips_per_step / (time.perf_counter() - t0)
        # the same stream as 16-bit PCM, the format wavfile_to_examples reads (vggish_input.py:85-99; the kernel applies the / 32768):
        # half the bytes over PCIe, so the copy hides under the compute
        host16 = torch.empty(pcm.shape, dtype=torch.int16).pin_memory()
        host16.copy_((pcm.clamp(-1, 1) * 32767).round().to(torch.int16))
        with torch.no_grad():
            for _ in ens.stream_waveforms([host16] * 2):
                pass
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for o in ens.stream_waveforms([host16] * n):
                pass
            torch.cuda.synchronize()
        out["clips_per_s_streamed_int16"] = n * clips_per_step / (time.perf_counter() - t0)
    return out


TRAIN_MFLOP_FWD = sum(CONV_MFLOP.values()) + sum(FC_MFLOP.values())       # per clip, conv + FC forward
# finetune: forward + weight gradient + input gradient of every layer (conv1 has no input gradient); the MLA head's
# ~5 MFLOP per clip are not counted
TRAIN_MFLOP = {"frozen": TRAIN_MFLOP_FWD, "finetune": 3 * TRAIN_MFLOP_FWD - CONV_MFLOP["conv1"]}


_TRAIN_SD = {}


def make_train_step(bags, finetune, precision, rank, device, sync_bn=True):
    W = importlib.import_module(PKG + ".weights")
    M = importlib.import_module(PKG + ".model")
    TR = importlib.import_module(PKG + ".train")
    ens = M.Ensemble("repeat", CNN_CONF, [2, 1], device, precision=precision)
    if not _TRAIN_SD:                                       # 73 M portable-seeded values: generated once per process
        _TRAIN_SD.update(W.make_state_dict(7, W.ensemble_shapes((2, 1), False)))
    ens.load_state_dict({k: torch.as_tensor(v) for k, v in _TRAIN_SD.items()})
    ens.to(device)
    if finetune:
        M.set_requires_grad(ens, True)
    step = TR.TrainStep(ens, lr=1e-4 if finetune else 1e-3, sync_bn=sync_bn)
    x = torch.from_numpy(W.uniform(4000 + rank, 1, bags * T_BAG * 96 * 64, lo=-1.4, hi=4.6)).reshape(bags, T_BAG, 1, 96, 64).to(device)
    y = torch.from_numpy(W.bits24(4000 + rank, 2, bags) % 10).to(device)
    return step, x, y


def train_leg(device, bags=512, steps=5, warmup=2):
    """BASELINE config 4 on the driver's clock (N = 1 line): the train.py step (zero_grad -> forward -> CrossEntropyLoss ->
    backward -> Adam) on 512 bags = 5 120 clips of 96 x 64 log-mel, frozen CNN in bf16 (the reference default
    cnn_trainable=False) and finetune (every parameter trainable, train.py:96-97) in bf16 with f32 master weights and in
    exact f32; FLOP-roofline fraction = algorithmic conv + FC flops (forward only when frozen; forward + dgrad + wgrad
    when finetuning) / step time / the MFMA peak of the arithmetic type."""
    out = {"bags": bags, "clips_per_step": bags * T_BAG}
    for name, finetune, precision, n in (("frozen_bf16", False, "bf16", steps), ("finetune_bf16", True, "bf16", steps),
                                         ("finetune_f32", True, "f32", 2)):
        try:
            step, x, y = make_train_step(bags, finetune, precision, 0, device)
        except NotImplementedError as e:
            out[name] = {"error": str(e)}
            continue
        for _ in range(warmup):
            loss, _ = step(x, y)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            loss, _ = step(x, y)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        tf = bags * T_BAG * TRAIN_MFLOP["finetune" if finetune else "frozen"] * 1e6 / dt / 1e12
        out[name] = {"ms_per_step": dt * 1e3, "clips_per_s": bags * T_BAG / dt, "TFLOPs": tf, "peak": PEAK_TFLOPS[precision],
                     "frac": tf / PEAK_TFLOPS[precision], "loss_last": float(loss)}
        del step, x, y
        torch.cuda.empty_cache()
    return out


def collective_summary(step, n_steps, world):
    """What the data-parallel exchange of the timed steps looked like, from the events ops.Dist / TrainStep recorded on the
    streams the collectives ran on (step.dist.trace, step.exposed): per step, averaged over the timed steps.
    bus_GBps = 2 (N - 1) / N x bytes / time, the per-link figure ring all-reduces are judged by (xGMI: 7 links x ~153 GB/s per GPU).
    overlap_hidden_frac = 1 - (time the compute stream waited for the communication stream) / (time of the gradient all-reduces)."""
    per = {}
    for tag, nbytes, e0, e1 in step.dist.trace:
        d = per.setdefault(tag, [0, 0, 0.0])
        d[0] += 1
        d[1] += nbytes
        d[2] += e0.elapsed_time(e1)
    grad = {k: v for k, v in per.items() if k.startswith("grad:")}
    bn = {k: v for k, v in per.items() if k.startswith("syncbn")}
    g_bytes = sum(v[1] for v in grad.values()) / n_steps
    g_ms = sum(v[2] for v in grad.values()) / n_steps
    bn_ms = sum(v[2] for v in bn.values()) / n_steps
    exposed_ms = sum(e0.elapsed_time(e1) for e0, e1 in (step.exposed or [])) / n_steps
    bucketed = bool(step.exposed)
    out = dict(step.dist.describe())
    out.update({
        "allreduce_bytes_per_step": g_bytes, "allreduce_ms": g_ms,
        "bus_GBps": (2.0 * (world - 1) / world * g_bytes / (g_ms * 1e-3) / 1e9) if g_ms > 0 else None,
        "gradient_messages_per_step": {k[5:]: {"bytes": v[1] // v[0], "ms": v[2] / v[0]} for k, v in grad.items()},
        "syncbn_allreduces_per_step": sum(v[0] for v in bn.values()) / n_steps, "syncbn_allreduce_ms": bn_ms,
        "syncbn_bytes_per_allreduce": (sum(v[1] for v in bn.values()) / max(sum(v[0] for v in bn.values()), 1)) if bn else None,
        "other_allreduces_per_step": sum(v[0] for k, v in per.items() if k in ("loss", "hits", "other")) / n_steps,
        "gradient_exchange": "buckets on a second HIP stream under the CNN backward" if bucketed else "one message on the compute stream",
        "exposed_wait_ms": exposed_ms if bucketed else g_ms,
        "overlap_hidden_frac": max(0.0, min(1.0, 1.0 - exposed_ms / g_ms)) if bucketed and g_ms > 0 else 0.0,
    })
    return out


DP_VARIANTS = (("frozen_bf16", False, True), ("frozen_bf16_per_shard_bn", False, False), ("finetune_bf16", True, True))


def dp_train_leg(world, rank, device, backend, bags=512, steps=5, warmup=2, variants=DP_VARIANTS):
    """BASELINE config 5 inside the line the driver's scaling run produces (`python bench.py --gpus N`, N > 1): after the
    inference headline EVERY rank runs the data-parallel train.py step (train.py:119-142: zero_grad -> forward -> CrossEntropyLoss
    -> backward -> Adam) on its own `bags` bags of 96 x 64 log-mel -- global batch N x bags, 4096 at N = 8 -- with SyncBN sums and
    the flat gradient buffer all-reduced over the group (RCCL through mla_allreduce_flat under backend nccl). Same barrier /
    max-over-ranks timing as the headline. Returns (train_step, collective) sections; rank 0 prints them."""
    import torch.distributed as dist
    ops = importlib.import_module(PKG + ".ops")

    def barrier():
        dist.barrier()
        torch.cuda.synchronize()

    out = {"bags_per_gpu": bags, "global_batch_bags": world * bags, "clips_per_step": world * bags * T_BAG, "n_gpus": world,
           "steps": steps, "warmup": warmup}
    coll = {}
    for name, finetune, sync_bn in variants:
        step, x, y = make_train_step(bags, finetune, "bf16", rank, device, sync_bn=sync_bn)
        for _ in range(warmup):
            loss, _ = step(x, y)
        torch.cuda.synchronize()
        step.dist.trace, step.exposed = [], []
        loss, _ = step(x, y)                                   # one traced step to learn how many events a step needs
        torch.cuda.synchronize()
        per_step = 2 * len(step.dist.trace) + 2 * len(step.exposed)
        step.dist.trace, step.exposed = [], []
        ops.reserve_events(per_step * steps + 16)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss, hits = step(x, y)
        barrier()
        elapsed = time.perf_counter() - t0
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        kind = "finetune" if finetune else "frozen"
        clips = world * bags * T_BAG
        tf = clips * TRAIN_MFLOP[kind] * 1e6 * steps / elapsed / 1e12
        assert bool(torch.isfinite(loss)), "non-finite loss in the data-parallel step"
        out[name] = {"ms_per_step": elapsed / steps * 1e3, "clips_per_s": clips * steps / elapsed, "TFLOPs": tf,
                     "peak": world * PEAK_TFLOPS["bf16"], "frac": tf / (world * PEAK_TFLOPS["bf16"]), "loss_last": float(loss),
                     "hits_last_global": int(hits[0]), "trainable_floats": step.n_params, "sync_bn": sync_bn}
        coll[name] = collective_summary(step, steps, world)
        step.dist.trace = step.exposed = None
        step.dist.close()
        del step, x, y
        torch.cuda.empty_cache()
    return out, coll


class Watchdog:
    """Hard deadline for a leg that runs collectives nobody has executed on this machine before: if it has not finished after
    `seconds`, `on_fire()` runs on a timer thread (rank 0 prints the line it has) and the process exits. A rank stuck in a
    collective never returns to Python, so nothing softer works; every rank arms the same deadline, so all of them leave."""

    def __init__(self, seconds, on_fire):
        import threading
        self.t = threading.Timer(seconds, self._fire)
        self.t.daemon = True
        self.on_fire = on_fire

    def _fire(self):
        try:
            self.on_fire()
        finally:
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(0)

    def __enter__(self):
        self.t.start()
        return self

    def __exit__(self, *exc):
        self.t.cancel()
        return False


def train_mode(args, world, rank, device):
    """BASELINE configs 4 / 5: one step = zero_grad -> forward -> CrossEntropyLoss -> backward -> Adam (train.py:124-138)
    on `bags` bags of 96 x 64 log-mel input per GPU (default 512; global batch = world x bags, 4096 at 8 GPUs), sharded by
    bag. Per step the ranks exchange the BatchNorm sums (SyncBN: the step equals the reference's single-process step
    on the global batch) and the flat gradient buffer over RCCL. Frozen CNN (the reference default) unless
    --finetune; CNN precision from --precision."""
    import torch.distributed as dist
    bags = args.bags if args.bags != 1024 else 512
    precision = args.precision
    step, x, y = make_train_step(bags, args.finetune, precision, rank, device, sync_bn=not args.no_sync_bn)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss, _ = step(x, y)
    if world > 1:                                           # collective evidence: events on the streams the all-reduces run on
        ops = importlib.import_module(PKG + ".ops")
        torch.cuda.synchronize()
        step.dist.trace, step.exposed = [], []
        step(x, y)
        torch.cuda.synchronize()
        per_step = 2 * len(step.dist.trace) + 2 * len(step.exposed)
        step.dist.trace, step.exposed = [], []
        ops.reserve_events(per_step * args.steps + 16)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = step(x, y)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    if rank == 0:
        clips = bags * T_BAG
        assert bool(torch.isfinite(loss))
        kind = "finetune" if args.finetune else "frozen"
        tf = world * clips * TRAIN_MFLOP[kind] * 1e6 * args.steps / elapsed / 1e12
        print(json.dumps({
            "metric": "0.96 s clips/sec train step fwd+bwd+Adam (VGGish+attn)", "value": world * clips * args.steps / elapsed,
            "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": precision, "data": "synthetic",
            "config": {"workload": "BASELINE config %s: train.py step on %d bags x 10 x (96 x 64) log-mel per GPU, %s, Adam lr %g, "
                                   "MLA [2,1]; SyncBN sums + flat-gradient all-reduce (%d floats) per step"
                                   % ("5" if world > 1 else "4", bags, "finetune (all parameters)" if args.finetune else "frozen CNN (reference default)",
                                      step.lr, step.n_params),
                       "bags_per_gpu": bags, "global_batch_bags": world * bags, "parallelism": "dp%d" % world},
            "roofline": {"bound": "mfma", "achieved": tf, "peak": world * PEAK_TFLOPS[precision], "unit": "TFLOP/s",
                         "frac": tf / (world * PEAK_TFLOPS[precision]), "traffic": None,
                         "flop_per_clip": TRAIN_MFLOP[kind] * 1e6, "what": "whole step, algorithmic conv + FC flops"},
            "collective": collective_summary(step, args.steps, world) if world > 1 else None,
            "loss_last": float(loss)}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv, timeout_s):
    """`python bench.py --gpus N` from a plain shell: spawn the N ranks (one process per GPU) as CHILD processes with the
    torch.distributed.run environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), relay rank 0's JSON line and return
    the first non-zero child exit code. The parent has made no GPU call when it gets here and makes none afterwards
    (a process that has initialised the GPU must never exec or fork workers). The library is built here, once, so that
    no rank imports a half-written .so."""
    import subprocess
    import tempfile
    importlib.import_module(PKG + ".build").build(verbose=False)
    port = free_port()
    base = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs, outs = [], []
    for r in range(n):
        out = tempfile.TemporaryFile(mode="w+") if r == 0 else subprocess.DEVNULL
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv,
                                      env=dict(base, RANK=str(r), LOCAL_RANK=str(r)), stdout=out))
    deadline, rc = time.time() + timeout_s, 0
    live = set(range(n))
    while live and rc == 0:
        for r in sorted(live):
            code = procs[r].poll()
            if code is not None:
                live.discard(r)
                if code != 0:
                    rc = code
                    sys.stderr.write("bench.py: rank %d exited with code %d\n" % (r, code))
        if time.time() > deadline:
            rc = 124
            sys.stderr.write("bench.py: ranks still running after %d s\n" % timeout_s)
        time.sleep(0.05)
    for r in live:                                    # a rank failed or timed out: stop exactly the children started here
        procs[r].terminate()
    for r in live:
        try:
            procs[r].wait(timeout=20)
        except subprocess.TimeoutExpired:
            procs[r].kill()
    outs[0].seek(0)
    sys.stdout.write(outs[0].read())
    sys.stdout.flush()
    return rc


def dry_run(args, world, rank):
    """Launch-path self-test without a GPU (tests/test_bench_launch_cpu.py): rendezvous, one all-reduce, one JSON line."""
    import torch.distributed as dist
    if rank == args.dry_run_fail_rank:
        raise SystemExit(7)
    total = float(rank)
    if world > 1:
        dist.init_process_group("gloo")
        t = torch.tensor([float(rank)], dtype=torch.float64)
        dist.all_reduce(t)
        total = float(t.item())
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "rank_sum": total, "mode": args.mode,
                          "legs": (["headline"] + ([] if args.no_train_leg else ["train_step", "collective"] if world > 1 else ["train_step"]))
                                  if args.mode == "infer" else ["train"],
                          "train_bags": args.train_bags, "train_steps": args.train_steps, "train_leg_timeout": args.train_leg_timeout,
                          "dp_variants": [v[0] for v in DP_VARIANTS]}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bags", type=int, default=1024, help="bags (10 s waveforms) per GPU per step")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f32", "bf16x3"],
                    help="conv/FC arithmetic: bf16 (BASELINE config 3), f32 = exact f32 MFMA, bf16x3 = three bf16 MFMA products per term (f32-grade)")
    ap.add_argument("--mode", default="infer", choices=["infer", "train"],
                    help="infer (default, the headline metric) or train: BASELINE configs 4/5, the data-parallel train.py step")
    ap.add_argument("--finetune", action="store_true", help="--mode train: every parameter trainable, train.py:96-97")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-small-batch", action="store_true", help="skip the 1 020-clip eager / HIP-graph leg")
    ap.add_argument("--no-parity-mode", action="store_true", help="skip the bf16x3 (1e-4-parity arithmetic) leg")
    ap.add_argument("--no-h2d", action="store_true", help="skip the host-buffer (PCIe-inclusive) leg")
    ap.add_argument("--no-train-leg", action="store_true", help="skip the training-step leg (config 4 at N = 1, the data-parallel config 5 step at N > 1)")
    ap.add_argument("--train-bags", type=int, default=512, help="bags per GPU of the training-step leg (BASELINE configs 4 / 5: 512)")
    ap.add_argument("--train-steps", type=int, default=5, help="timed steps per variant of the N > 1 training-step leg")
    ap.add_argument("--train-leg-timeout", type=int, default=420, help="seconds the N > 1 training-step leg may take before rank 0 prints the line without it")
    ap.add_argument("--no-sync-bn", action="store_true", help="--mode train: per-shard BatchNorm statistics (DDP semantics) instead of SyncBN")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo to rehearse N>1 on one GPU)")
    ap.add_argument("--launch-timeout", type=int, default=1500, help="seconds the self-launched ranks may run")
    ap.add_argument("--prewarm-seconds", type=float, default=1.5,
                    help="device warm-up before the W warm-up steps: the first seconds of work on an idle MI355X run at ramping clocks "
                         "(measured: the first process on a fresh box is 8-10 %% slower); untimed, like model construction")
    ap.add_argument("--dry-run", action="store_true", help="launch-path self-test: no GPU work (CPU tests)")
    ap.add_argument("--dry-run-fail-rank", type=int, default=-1, help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:], args.launch_timeout))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.dry_run:
        return dry_run(args, world, rank)
    local_rank %= max(torch.cuda.device_count(), 1)          # rehearsal on fewer GPUs than ranks (gloo only)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)
    # under torch.distributed.run nobody built the library for us: rank 0 does, everybody else waits for it
    if rank == 0:
        importlib.import_module(PKG + ".build").build(verbose=False)
    if world > 1:
        dist.barrier()
    ops = importlib.import_module(PKG + ".ops")
    if args.mode == "train":
        return train_mode(args, world, rank, device)
    return infer_mode(args, world, rank, device, ops)


def load_pmc_traffic():
    """HBM bytes per clip of the dominant kernels, from rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE in separate
    passes (profiles/r02_pmc_traffic.json, written by scripts/profile_traffic.sh). PMC counters cannot be read from inside
    this process; per-launch traffic = per-clip figure x the clips of this launch (both kernels stream per clip)."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))       # the newest round's file
    if not paths:
        return {}
    with open(paths[-1]) as f:
        d = json.load(f)
    d["_file"] = os.path.basename(paths[-1])
    return d


PEAK_CLOCK_GHZ, N_SIMD, N_CU = 2.4, 1024, 256        # MI355X_MICROARCH.md: max clock, 256 CUs x 4 SIMD-32


def frontend_issue_bound(t_fe, clips, launch_s):
    """The bound that actually binds the log-mel kernel (its HBM traffic is 1.00x the algorithmic bytes at 0.3 of the HBM peak):
    vector-instruction issue and the LDS array. From the PMC pass on file (scripts/profile_traffic.sh, third pass):
    wave-instructions per clip x clips of this launch. A SIMD-32 issues one wave64 vector instruction per 2 cycles at best
    (MI355X_MICROARCH.md, wave scheduling; transcendentals take longer, so this is a LOWER bound on the time); the LDS array
    serves one access cycle per CU and clock. Both priced at the 2.4 GHz peak clock; `frac` = bound / measured launch time."""
    if not t_fe or "issue" not in t_fe:
        return None
    i = t_fe["issue"]
    valu_s = i["valu_insts_per_clip"] * clips * 2.0 / (N_SIMD * PEAK_CLOCK_GHZ * 1e9)
    lds_s = i["lds_array_cycles_per_clip"] * clips / (N_CU * PEAK_CLOCK_GHZ * 1e9)
    return {"valu_insts_per_clip": i["valu_insts_per_clip"], "valu_bound_ms": valu_s * 1e3, "frac_of_valu_bound": valu_s / launch_s,
            "lds_array_cycles_per_clip": i["lds_array_cycles_per_clip"], "lds_bound_ms": lds_s * 1e3, "frac_of_lds_bound": lds_s / launch_s,
            "lds_conflict_share": i["lds_conflict_cycles_per_clip"] / i["lds_array_cycles_per_clip"] if i["lds_array_cycles_per_clip"] else None,
            "clock_GHz_while_profiled": i.get("clock_GHz_while_profiled"), "source": i["source"]}


def infer_mode(args, world, rank, device, ops):
    import torch.distributed as dist
    ens, sd = build_model(args.precision, device)
    pcm = synth_pcm(args.bags, rank, device)
    clips_per_step = args.bags * T_BAG

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < args.prewarm_seconds:     # clocks up before anything is measured (not part of W or K)
            ens.forward_waveforms(pcm)
            torch.cuda.synchronize()
        for _ in range(args.warmup):
            out = ens.forward_waveforms(pcm)
        assert tuple(out.shape) == (args.bags, 10) and bool(torch.isfinite(out).all())
        # the per-kernel HIP events of the timed region exist before it starts (creating one inside can stall the launching thread)
        ops.profile = []
        ens.forward_waveforms(pcm)
        per_step, ops.profile = len(ops.profile), None
        ops.reserve_events(2 * per_step * args.steps + 16)
        barrier()
        ops.profile = []
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = ens.forward_waveforms(pcm)
        barrier()
        elapsed = time.perf_counter() - t0
        prof, ops.profile = ops.profile, None

    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        avg = kernel_averages(prof)
        dom = max((k for k in avg if k in CONV_DESC), key=lambda k: avg[k])
        tf = clips_per_step * CONV_MFLOP[dom] * 1e6 / avg[dom] / 1e12
        peak = PEAK_TFLOPS[args.precision]
        conv_t = sum(avg[k] for k in avg if k in CONV_MFLOP)
        conv_tf = clips_per_step * sum(CONV_MFLOP.values()) * 1e6 / conv_t / 1e12
        fe_dtype = torch.bfloat16 if args.precision == "bf16" else torch.float32
        fe_gbps = clips_per_step * FE_BYTES[fe_dtype] / avg["logmel"] / 1e9
        traffic = load_pmc_traffic()
        t_dom = traffic.get("%s/%s" % (dom, args.precision))
        t_fe = traffic.get("logmel/%s" % ("bf16" if fe_dtype == torch.bfloat16 else "f32"))
        result = {
            "metric": "0.96 s clips/sec wave->logits (VGGish+attn)", "value": world * clips_per_step * args.steps / elapsed,
            "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "BASELINE config 3: wave->logits forward, %d bags x 10 s 16 kHz PCM per GPU = %d x 0.96 s clips "
                                   "per step, VGGish + MLA [2,1], random-init portable-seeded weights; f32 front-end/head, "
                                   "%s conv+FC" % (args.bags, clips_per_step, args.precision),
                       "bags_per_gpu": args.bags, "clips_per_step_per_gpu": clips_per_step, "parallelism": "dp%d (no collective: independent bags)" % world},
            "roofline": {"bound": "mfma", "kernel": "conv3x3_kernel %s (%s)" % (dom, CONV_DESC[dom]), "achieved": tf, "peak": peak,
                         "unit": "TFLOP/s", "frac": tf / peak,
                         "traffic": t_dom["bytes_per_clip"] * clips_per_step if t_dom else None,
                         "traffic_source": t_dom["source"] if t_dom else "no PMC pass on file for this kernel / dtype",
                         "algorithmic_bytes": CONV_BYTES[dom](2 if args.precision == "bf16" else 4) * clips_per_step if dom in CONV_BYTES else None,
                         "avg_launch_ms": avg[dom] * 1e3, "flop_per_launch": clips_per_step * CONV_MFLOP[dom] * 1e6},
            "roofline_conv_stack": {"bound": "mfma", "achieved": conv_tf, "peak": peak, "unit": "TFLOP/s", "frac": conv_tf / peak,
                                    "ms": conv_t * 1e3,
                                    "per_kernel_frac": {k: round(clips_per_step * m * 1e6 / avg[k] / 1e12 / peak, 4)
                                                        for k, m in list(CONV_MFLOP.items()) + list(FC_MFLOP.items()) if k in avg}},
            "roofline_frontend": {"bound": "hbm", "kernel": "logmel_dyn_kernel", "achieved": fe_gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                                  "frac": fe_gbps / PEAK_HBM_GBPS, "avg_launch_ms": avg["logmel"] * 1e3,
                                  "bytes_per_clip": FE_BYTES[fe_dtype],
                                  "traffic": t_fe["bytes_per_clip"] * clips_per_step if t_fe else None,
                                  "issue": frontend_issue_bound(t_fe, clips_per_step, avg["logmel"])},
            "kernel_ms": {k: round(v * 1e3, 4) for k, v in sorted(avg.items())},
        }
        def leg(name, fn):
            """The additional legs of the N = 1 line never cost the headline line: a failure is reported in place."""
            try:
                result[name] = fn()
            except Exception as e:                                   # noqa: BLE001 -- reported, not hidden
                result[name] = {"error": "%s: %s" % (type(e).__name__, e)}
                print("bench.py: leg %s failed: %r" % (name, e), file=sys.stderr, flush=True)

        if world == 1 and args.precision == "bf16" and not args.no_parity_mode:
            leg("parity_mode", lambda: parity_mode_leg(ens, pcm, clips_per_step, ops))
        if world == 1 and not args.no_h2d:
            leg("h2d", lambda: h2d_leg(pcm, elapsed / args.steps, clips_per_step, ens=ens))
        if world == 1 and not args.no_small_batch:
            leg("small_batch", lambda: small_batch_leg(ens, rank, device))
        if world == 1 and not args.no_cpu_baseline:
            def cpu_leg():
                with torch.no_grad():
                    return cpu_baseline(sd, ens, device)
            leg("cpu_baseline", cpu_leg)
        # north star: "logits within 1e-4 rel of the CPU reference" -- the throughput of the fastest arithmetic mode whose
        # MEASURED deviation from the oracle meets it, stated at top level beside `value` (which is BASELINE config 3's bf16)
        if "parity_max_rel_bf16" in result.get("cpu_baseline", {}) and "bf16x3" in result.get("parity_mode", {}):
            cands = [("bf16", result["value"], result["roofline_conv_stack"]["frac"], result["roofline_conv_stack"]["frac"])]
            for m in ("bf16x3", "f32"):
                pm = result["parity_mode"][m]
                cands.append((m, pm["clips_per_s"], pm.get("conv_stack_frac_algorithmic"), pm.get("conv_stack_frac_issued")))
            ok = [c for c in cands if result["cpu_baseline"]["parity_max_rel_%s" % c[0]] <= 1e-4]
            if ok:
                best = max(ok, key=lambda c: c[1])
                result["value_at_tolerance"] = {"value": best[1], "unit": "clips/s", "mode": best[0], "tolerance": 1e-4,
                                                "measured_max_rel": result["cpu_baseline"]["parity_max_rel_%s" % best[0]],
                                                "conv_stack_frac_algorithmic": best[2], "conv_stack_frac_issued": best[3]}
        if world == 1 and not args.no_train_leg:
            del ens, pcm
            torch.cuda.empty_cache()
            leg("train_step", lambda: train_leg(device, bags=args.train_bags))
        if world == 1 or args.no_train_leg:
            print(json.dumps(result), flush=True)
    if world > 1:
        if not args.no_train_leg:
            # BASELINE config 5 in the line the driver's scaling run produces: every rank runs the data-parallel train step
            state = {"printed": False}
            ens = pcm = None
            torch.cuda.empty_cache()

            def give_up():
                if rank == 0 and not state["printed"]:
                    result["train_step"] = {"error": "the data-parallel train leg did not finish within %d s (rank 0 gave up waiting)" % args.train_leg_timeout}
                    print(json.dumps(result), flush=True)
                sys.stderr.write("bench.py: rank %d: data-parallel train leg timed out\n" % rank)

            with Watchdog(args.train_leg_timeout, give_up):
                try:
                    ts, coll = dp_train_leg(world, rank, device, args.backend, bags=args.train_bags, steps=args.train_steps)
                    if rank == 0:
                        result["train_step"], result["collective"] = ts, coll
                except Exception as e:                               # noqa: BLE001 -- reported in the line, not hidden
                    if rank == 0:
                        result["train_step"] = {"error": "%s: %s" % (type(e).__name__, e)}
                    print("bench.py: rank %d: data-parallel train leg failed: %r" % (rank, e), file=sys.stderr, flush=True)
                if rank == 0:
                    print(json.dumps(result), flush=True)
                    state["printed"] = True
                dist.barrier()
                dist.destroy_process_group()
            return
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
