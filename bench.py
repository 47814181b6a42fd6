#!/usr/bin/env python3
"""Headline benchmark: 0.96 s clips/s, wave -> class scores (VGGish + multi-level attention).

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)
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
# HBM bytes per launch of the dominant kernel at the default batch, from rocprofv3 --pmc FETCH_SIZE (x2, gfx950)
# + WRITE_SIZE in separate passes: profiles/r01_pmc_traffic.txt. Not measurable from inside this process.
PMC_TRAFFIC = {("conv4", "bf16", 10240): 3.38e9}
PEAK_HBM_GBPS = 8000.0

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
    """BASELINE config 3 read literally ("batch 1024 clips"): 102 bags = 1 020 clips per step, where one step is
    ~1.5 ms of GPU work behind ~45 launches. Timed eagerly and as one HIP-graph replay per step."""
    pcm = synth_pcm(bags, rank, device)
    out = {"clips_per_step": bags * T_BAG}
    with torch.no_grad():
        g = ens.capture_waveforms(pcm)
        ref = ens.forward_waveforms(pcm).clone()
        assert torch.equal(g(pcm), ref), "graph replay must reproduce the eager result"
        for name, fn in (("eager", lambda: ens.forward_waveforms(pcm)), ("hip_graph", lambda: g(pcm))):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            out[name] = {"ms_per_step": dt * 1e3, "clips_per_s": bags * T_BAG / dt}
    return out


def parity_mode_leg(ens, pcm, clips_per_step, steps=5):
    """The same batch in the bf16x3 mode: still bf16 MFMA arithmetic, but every value carried as hi + lo bf16 planes and
    every product as three bf16 terms with f32 accumulation, which meets the north star's 1e-4 relative tolerance on the
    scores (the plain bf16 headline does not: `cpu_baseline.parity_max_rel_*`). Also the exact-f32 MFMA mode beside it."""
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
            out[prec] = {"ms_per_step": dt * 1e3, "clips_per_s": clips_per_step / dt}
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
    if ens is not None:            # measured: Ensemble.stream_waveforms (copy stream + two device buffers) over 8 host batches
        n = 8
        with torch.no_grad():
            for _ in ens.stream_waveforms([host] * 2):
                pass
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for o in ens.stream_waveforms([host] * n):
                pass
            torch.cuda.synchronize()
        out["clips_per_s_streamed"] = n * clips_per_step / (time.perf_counter() - t0)
    return out


def train_mode(args, world, rank, device):
    """BASELINE configs 4 / 5: one step = zero_grad -> forward -> CrossEntropyLoss -> backward -> Adam (train.py:124-138)
    on `bags` bags of 96 x 64 log-mel input per GPU (default 512; global batch = world x bags, 4096 at 8 GPUs), sharded by
    bag. Per step the ranks exchange the BatchNorm sums (SyncBN: the step equals the reference's single-process step
    on the global batch) and ONE flat-gradient all-reduce over RCCL. Frozen CNN (the reference default) unless
    --finetune; CNN precision from --precision (finetune: f32)."""
    import torch.distributed as dist
    W = importlib.import_module(PKG + ".weights")
    M = importlib.import_module(PKG + ".model")
    TR = importlib.import_module(PKG + ".train")
    bags = args.bags if args.bags != 1024 else 512
    precision = "f32" if args.finetune else args.precision
    ens = M.Ensemble("repeat", CNN_CONF, [2, 1], device, precision=precision)
    ens.load_state_dict({k: torch.as_tensor(v) for k, v in W.make_state_dict(7, W.ensemble_shapes((2, 1), False)).items()})
    ens.to(device)
    if args.finetune:
        M.set_requires_grad(ens, True)
    step = TR.TrainStep(ens, lr=1e-4 if args.finetune else 1e-3)
    x = torch.from_numpy(W.uniform(4000 + rank, 1, bags * T_BAG * 96 * 64, lo=-1.4, hi=4.6)).reshape(bags, T_BAG, 1, 96, 64).to(device)
    y = torch.from_numpy(W.bits24(4000 + rank, 2, bags) % 10).to(device)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss, _ = step(x, y)
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
        print(json.dumps({
            "metric": "0.96 s clips/sec train step fwd+bwd+Adam (VGGish+attn)", "value": world * clips * args.steps / elapsed,
            "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": precision, "data": "synthetic",
            "config": {"workload": "BASELINE config %s: train.py step on %d bags x 10 x (96 x 64) log-mel per GPU, %s, Adam lr %g, "
                                   "MLA [2,1]; SyncBN sums + one flat-gradient all-reduce (%d floats) per step"
                                   % ("5" if world > 1 else "4", bags, "finetune (all parameters)" if args.finetune else "frozen CNN (reference default)",
                                      step.lr, step.n_params),
                       "bags_per_gpu": bags, "global_batch_bags": world * bags, "parallelism": "dp%d" % world},
            "loss_last": float(loss)}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


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
    ap.add_argument("--finetune", action="store_true", help="--mode train: every parameter trainable (f32), train.py:96-97")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-small-batch", action="store_true", help="skip the 1 020-clip eager / HIP-graph leg")
    ap.add_argument("--no-parity-mode", action="store_true", help="skip the bf16x3 (1e-4-parity arithmetic) leg")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo to rehearse N>1 on one GPU)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..." % (args.gpus, args.gpus))
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

    importlib.import_module(PKG + ".build").build(verbose=False) if rank == 0 and not os.path.exists(
        os.path.join(ROOT, PKG, "libmla_hip.so")) else None
    ops = importlib.import_module(PKG + ".ops")
    if args.mode == "train":
        return train_mode(args, world, rank, device)
    ens, sd = build_model(args.precision, device)
    pcm = synth_pcm(args.bags, rank, device)
    clips_per_step = args.bags * T_BAG

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            out = ens.forward_waveforms(pcm)
        assert tuple(out.shape) == (args.bags, 10) and bool(torch.isfinite(out).all())
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
        per = {}
        for name, e0, e1 in prof:
            per.setdefault(name, []).append(e0.elapsed_time(e1) * 1e-3)
        avg = {k: sum(v) / len(v) for k, v in per.items()}
        dom = max((k for k in avg if k in CONV_DESC), key=lambda k: avg[k])
        tf = clips_per_step * CONV_MFLOP[dom] * 1e6 / avg[dom] / 1e12
        peak = PEAK_TFLOPS[args.precision]
        conv_t = sum(avg[k] for k in avg if k in CONV_MFLOP)
        conv_tf = clips_per_step * sum(CONV_MFLOP.values()) * 1e6 / conv_t / 1e12
        fe_dtype = torch.bfloat16 if args.precision == "bf16" else torch.float32
        fe_gbps = clips_per_step * FE_BYTES[fe_dtype] / avg["logmel"] / 1e9
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
                         "unit": "TFLOP/s", "frac": tf / peak, "traffic": PMC_TRAFFIC.get((dom, args.precision, clips_per_step)),
                         "avg_launch_ms": avg[dom] * 1e3, "flop_per_launch": clips_per_step * CONV_MFLOP[dom] * 1e6},
            "roofline_conv_stack": {"bound": "mfma", "achieved": conv_tf, "peak": peak, "unit": "TFLOP/s", "frac": conv_tf / peak,
                                    "ms": conv_t * 1e3},
            "roofline_frontend": {"bound": "hbm", "kernel": "logmel_kernel", "achieved": fe_gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                                  "frac": fe_gbps / PEAK_HBM_GBPS, "avg_launch_ms": avg["logmel"] * 1e3,
                                  "bytes_per_clip": FE_BYTES[fe_dtype]},
            "kernel_ms": {k: round(v * 1e3, 4) for k, v in sorted(avg.items())},
        }
        if world == 1 and args.precision == "bf16" and not args.no_parity_mode:
            result["parity_mode"] = parity_mode_leg(ens, pcm, clips_per_step)
        if world == 1:
            result["h2d"] = h2d_leg(pcm, elapsed / args.steps, clips_per_step, ens=ens)
        if world == 1 and not args.no_small_batch:
            result["small_batch"] = small_batch_leg(ens, rank, device)
        if world == 1 and not args.no_cpu_baseline:
            with torch.no_grad():
                result["cpu_baseline"] = cpu_baseline(sd, ens, device)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
