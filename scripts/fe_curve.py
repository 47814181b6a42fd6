"""Front-end time against batch size (f32 PCM in, bf16 out): per-waveform cost falls with the launch size.
    python scripts/fe_curve.py        (MLA_LOGMEL_SYNC=static selects the round-1 static kernel for A/B)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import fe_bench as fb
for n in (64, 128, 256, 512, 1024, 1536, 2048, 4096):
    r = fb.run(n, 160000, torch.float32, torch.bfloat16, iters=30)
    print(n, round(r["ms"], 4), "ms", round(r["ms"] / n * 1e3, 4), "us/waveform", round(r["frac_hbm"], 4), "of 8 TB/s")
