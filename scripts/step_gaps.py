"""Where does wall time go between steps? CPU enqueue time per step and GPU time per step (events at step boundaries) of the bench
pipeline, 20 steps after the usual warm-up.   python scripts/step_gaps.py"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
importlib.import_module(bench.PKG + ".build").build(verbose=False)
ens, sd = bench.build_model("bf16", dev)
pcm = bench.synth_pcm(1024, 0, dev)
with torch.no_grad():
    t = time.perf_counter()
    while time.perf_counter() - t < 1.5:
        ens.forward_waveforms(pcm); torch.cuda.synchronize()
    for _ in range(3):
        ens.forward_waveforms(pcm)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
    cpu = []
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(20):
        a = time.perf_counter()
        ens.forward_waveforms(pcm)
        ev[i + 1].record()
        cpu.append((time.perf_counter() - a) * 1e3)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
gpu = [ev[i].elapsed_time(ev[i + 1]) for i in range(20)]
print("wall %.1f ms for 20 steps = %.2f ms/step" % (wall, wall / 20))
ops = importlib.import_module(bench.PKG + ".ops")
with torch.no_grad():
    ops.profile = []
    cpu2 = []
    t0 = time.perf_counter()
    for i in range(20):
        a = time.perf_counter()
        ens.forward_waveforms(pcm)
        cpu2.append((time.perf_counter() - a) * 1e3)
    torch.cuda.synchronize()
    wall2 = (time.perf_counter() - t0) * 1e3
    n_ev = len(ops.profile)
    ops.profile = None
print("with per-kernel events (%d pairs): wall %.2f ms/step; cpu enqueue ms/step: %s" % (n_ev, wall2 / 20, " ".join("%.1f" % c for c in cpu2)))
print("cpu enqueue ms/step:", " ".join("%.1f" % c for c in cpu))
print("gpu ms/step        :", " ".join("%.1f" % g for g in gpu))
print("threads: torch %d, cpu_count %d, sched_affinity %d" % (torch.get_num_threads(), os.cpu_count(), len(os.sched_getaffinity(0))))
