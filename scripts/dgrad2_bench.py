"""dgrad of conv2 (128 -> 64 @48x32, bf16, 5 120 images): time and a checksum, for A/B builds (MLA_HIP_LIB)."""
import importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
PKG = "audio-classification-using-a-deep-cnn-combined-with-multi-level-attention_amd"
ops = importlib.import_module(PKG + ".ops")
W = importlib.import_module(PKG + ".weights")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5120
dev = torch.device("cuda", 0)
base = torch.from_numpy(W.uniform(7, 1, 64 * 48 * 32 * 128, lo=-1, hi=1)).reshape(64, 48, 32, 128)
dz = ops.to_bf16(base.repeat((n + 63) // 64, 1, 1, 1)[:n].contiguous().to(dev))
w = (torch.from_numpy(W.uniform(7, 2, 128 * 64 * 9)).reshape(128, 64, 3, 3) * 0.05).to(dev)
wd = ops.repack_dgrad(w, torch.bfloat16)
for _ in range(3):
    out = ops.conv3x3(dz, wd, None, 64, pool=False, act=False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    out = ops.conv3x3(dz, wd, None, 64, pool=False, act=False)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(json.dumps({"layer": "dgrad2", "n": n, "ms": ms, "TFLOPs": 2.0 * n * 48 * 32 * 64 * 9 * 128 / ms / 1e9, "checksum": float(out.float().double().sum()),
                  "abs": float(out.float().double().abs().sum())}))
