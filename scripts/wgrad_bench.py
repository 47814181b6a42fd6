"""wgrad micro-benchmark (bf16): the five VGGish weight-gradient shapes at 5 120 clips per launch."""
import importlib, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
PKG = "audio-classification-using-a-deep-cnn-combined-with-multi-level-attention_amd"
ops = importlib.import_module(PKG + ".ops")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5120
dt = torch.bfloat16 if (len(sys.argv) < 3 or sys.argv[2] == "bf16") else torch.float32
for layer, (cin, cout, H, W) in {2: (64, 128, 48, 32), 3: (128, 256, 24, 16), 4: (256, 256, 24, 16), 5: (256, 512, 12, 8), 6: (512, 512, 12, 8)}.items():
    a = (torch.rand((n, H, W, cin), device="cuda") * 2 - 0.5).clamp_min(0).to(dt)
    dz = ((torch.rand((n, H, W, cout), device="cuda") - 0.5) * (torch.rand((n, H, W, cout), device="cuda") > 0.5)).to(dt)
    dw = torch.empty((cout, cin, 3, 3), device="cuda")
    for _ in range(2):
        ops.conv_wgrad(dz, a, dw)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    ev[0].record()
    for i in range(5):
        ops.conv_wgrad(dz, a, dw)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(5))[2]
    print(json.dumps({"layer": layer, "n": n, "dtype": str(dt), "ms": ms, "TFLOPs": 2.0 * n * H * W * 9 * cin * cout / ms / 1e9}))
