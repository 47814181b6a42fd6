"""Calibration: how long a plain fill / copy of conv1's 2 GB output takes on this chip (the store-bound floor of conv1)."""
import torch, time
x = torch.empty(10240*48*32*64, dtype=torch.bfloat16, device="cuda")
y = torch.empty_like(x)
for f, name in ((lambda: x.zero_(), "zero_"), (lambda: x.fill_(1.0), "fill_"), (lambda: y.copy_(x), "copy_ (read+write)")):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(name, ms, "ms", x.numel() * 2 / ms / 1e6, "GB/s written")
