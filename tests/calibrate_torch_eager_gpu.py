"""Calibration only (test infrastructure, lives under tests/ because it uses the oracle; never on the product path): the oracle's torch restatement of the reference model run by PyTorch-ROCm
eager on the GPU (MIOpen convolutions, hipBLASLt/rocBLAS linears) -- i.e. what the reference's own model code costs on
this hardware, examples -> scores, f32 and bf16 autocast. First calls include MIOpen's kernel search (minutes)."""
import importlib, os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
PKG = "audio-classification-using-a-deep-cnn-combined-with-multi-level-attention_amd"
W = importlib.import_module(PKG + ".weights")
from oracle import model as omodel
sd = {k: torch.as_tensor(v).cuda() for k, v in W.make_state_dict(6, W.ensemble_shapes((2, 1), False)).items()}
bags = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
x = torch.from_numpy(W.uniform(1, 1, bags * 10 * 96 * 64, lo=-1.4, hi=4.6)).reshape(bags, 10, 1, 96, 64).cuda()
for name, ctx in (("f32", torch.autocast("cuda", enabled=False)), ("bf16 autocast", torch.autocast("cuda", dtype=torch.bfloat16))):
    with torch.no_grad(), ctx:
        for _ in range(2):
            out = omodel.ensemble_forward(sd, x, (2, 1), False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 3
        for _ in range(n):
            out = omodel.ensemble_forward(sd, x, (2, 1), False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
    print(json.dumps({"torch_eager": name, "bags": bags, "ms": dt * 1e3, "clips_per_s": bags * 10 / dt}))
