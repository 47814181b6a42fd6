"""conv1 micro-benchmark: the first VGGish layer alone (bf16 / f32 output) on a 10 240-clip batch, HIP-event timing."""
import importlib, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
PKG = "audio-classification-using-a-deep-cnn-combined-with-multi-level-attention_amd"
ops = importlib.import_module(PKG + ".ops")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10240
x = (torch.rand((n, 96, 64), device="cuda") * 6 - 1.4).to(torch.bfloat16)
w = (torch.rand((64, 1, 3, 3), device="cuda") - 0.5); b = torch.zeros(64, device="cuda")
for _ in range(3): ops.conv1(x, w, b, torch.bfloat16)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ops.conv1(x, w, b, torch.bfloat16)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(json.dumps({"n": n, "ms": ms, "GBps_out": n * 48 * 32 * 64 * 2 / ms / 1e6}))
