"""Conv-stack micro-benchmark: each VGGish conv layer alone on a 10 240-clip batch (bf16 or f32),
HIP-event timing; also the target of the rocprofv3 --pmc passes."""
import importlib, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
PKG = "audio-classification-using-a-deep-cnn-combined-with-multi-level-attention_amd"
ops = importlib.import_module(PKG + ".ops")
W = importlib.import_module(PKG + ".weights")
MFLOP = {2: 226.49, 3: 226.49, 4: 452.98, 5: 226.49, 6: 452.98}

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10240
    dtype = torch.bfloat16 if (len(sys.argv) < 3 or sys.argv[2] == "bf16") else torch.float32
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    layers = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [2, 3, 4, 5, 6]
    for layer in layers:
        (h, w, cin), (ho, wo, cout) = ops.CONV_SHAPES[layer]
        x = (torch.rand((n, h, w, cin), device="cuda") * 2 - 0.5).to(dtype)
        wt = torch.from_numpy(W.uniform(1, layer, cout * cin * 9)).reshape(cout, cin, 3, 3).cuda() * (6.0 / (9 * cin)) ** 0.5
        b = torch.zeros(cout, device="cuda")
        wp = ops.repack_conv_weight(wt, dtype)
        for _ in range(2):
            ops.conv(layer, x, wp, b)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
        ev[0].record()
        for i in range(iters):
            ops.conv(layer, x, wp, b)
            ev[i + 1].record()
        torch.cuda.synchronize()
        ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(iters))
        med = ts[len(ts) // 2] * 1e-3
        print(json.dumps({"layer": layer, "n": n, "dtype": str(dtype), "ms": med * 1e3, "TFLOPs": n * MFLOP[layer] * 1e6 / med / 1e12}))

if __name__ == "__main__":
    main()
