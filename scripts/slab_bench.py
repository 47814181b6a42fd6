"""Does running the pipeline slab by slab (so that each layer's output is still in the 256 MiB Infinity Cache when the next
layer reads it) beat layer-by-layer over the whole batch? 1024 bags, bf16; slab = bags per pass."""
import importlib, os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
ens, sd = bench.build_model("bf16", dev)
pcm = bench.synth_pcm(1024, 0, dev)
T = 10
def run(slab):
    outs = []
    for i in range(0, 1024, slab):
        outs.append(ens.forward_waveforms(pcm[i:i + slab]))
    return torch.cat(outs)
with torch.no_grad():
    ref = run(1024)
    for rnd in range(2):
        for slab in (1024, 512, 256, 128, 96, 64, 48, 32):
            out = run(slab)
            assert torch.equal(out, ref)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                run(slab)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 5
            print(json.dumps({"slab_bags": slab, "clips_per_pass": slab * 10, "ms": dt * 1e3, "clips_per_s": 10240 / dt}))
