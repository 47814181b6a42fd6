"""Target for rocprofv3 / timing: the FC1 / FC2 / FC3 shapes at a small batch (default 1 020 rows), bf16, 30 launches each.
    python scripts/gemm_small_prof.py [rows]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
PKG = "audio-classification-using-a-deep-cnn-combined-with-multi-level-attention_amd"
ops = importlib.import_module(PKG + ".ops")
W = importlib.import_module(PKG + ".weights")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1020
dev = torch.device("cuda", 0)
for K, N in ((12288, 4096), (4096, 4096), (4096, 128)):
    a = ops.to_bf16(torch.from_numpy(W.uniform(5, 1, M * K)).reshape(M, K).to(dev).clamp_min(0))
    w = ops.to_bf16((torch.from_numpy(W.uniform(5, 2, N * K)).reshape(N, K) * 0.02).to(dev))
    b = torch.zeros(N, device=dev)
    for _ in range(5):
        ops.linear(a, w, b, relu=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        ops.linear(a, w, b, relu=True)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    print("M %d K %d N %d: %.1f us, %.0f TFLOP/s, L2->LDS %.1f GB/s per CU (128 x 128 tiles)" % (M, K, N, us, 2.0 * M * N * K / us / 1e6,
          ((M + 127) // 128) * ((N + 127) // 128) * 256 * K * 2 / us / 1e3 / 256))
