"""Per-kernel time of the wave -> scores forward at a given number of bags (default 102 = BASELINE config 3 read literally,
1 020 clips), from events on the launch stream, next to the same kernels' per-clip time at 1 024 bags.
    python scripts/small_batch_prof.py [bags]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
ops = importlib.import_module(bench.PKG + ".ops")
dev = torch.device("cuda", 0)
ens, _ = bench.build_model("bf16", dev)
res = {}
for bags in (int(sys.argv[1]) if len(sys.argv) > 1 else 102, 1024):
    pcm = bench.synth_pcm(bags, 0, dev)
    with torch.no_grad():
        for _ in range(10):
            ens.forward_waveforms(pcm)
        torch.cuda.synchronize()
        ops.profile = []
        for _ in range(20):
            ens.forward_waveforms(pcm)
        torch.cuda.synchronize()
        res[bags] = bench.kernel_averages(ops.profile)
        ops.profile = None
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(50):
            ens.forward_waveforms(pcm)
        t1.record(); torch.cuda.synchronize()
        print("%d bags: %.3f ms per step (device span), sum of kernels %.3f ms" % (bags, t0.elapsed_time(t1) / 50, 1e3 * sum(res[bags].values())))
small, big = sorted(res)
print("%-22s %10s %10s %12s" % ("kernel", "us @%d" % small, "us @%d" % big, "per-clip ratio"))
for k in sorted(res[small], key=lambda k: -res[small][k]):
    a, b = res[small][k] * 1e6, res[big][k] * 1e6
    print("%-22s %10.1f %10.1f %12.2f" % (k, a, b, (a / small) / (b / big)))
