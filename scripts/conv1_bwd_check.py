"""conv1 backward (bf16 gradient in): error against torch autograd on the bf16-rounded operands, time per call at 5 120 clips,
determinism.   python scripts/conv1_bwd_check.py   (MLA_HIP_LIB=<variant built with -DMLA_CONV1_BWD_MFMA=0> for the vector-pipe kernel)"""
import importlib, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
PKG = "audio-classification-using-a-deep-cnn-combined-with-multi-level-attention_amd"
ops = importlib.import_module(PKG + ".ops")
torch.manual_seed(0)
n = 512
x = (torch.rand((n, 96, 64), device="cuda") * 6 - 1.4)
w = (torch.rand((64, 1, 3, 3), device="cuda") - 0.5) * 0.6
b = (torch.rand(64, device="cuda") - 0.5) * 0.2
d = ((torch.rand((n, 48, 32, 64), device="cuda") - 0.5)).to(torch.bfloat16)
dw = torch.empty((64, 1, 3, 3), device="cuda"); db = torch.empty(64, device="cuda")
ops.conv1_bwd(x, w, b, d, dw, db)
torch.cuda.synchronize()
# torch reference with the forward's bf16 operands (x, w rounded to bf16; f32 accumulate)
xr = x.to(torch.bfloat16).float().cpu()[:, None].requires_grad_(False)
wr = w.to(torch.bfloat16).float().cpu().requires_grad_(True)
br = b.cpu().clone().requires_grad_(True)
y = torch.nn.functional.max_pool2d(torch.relu(torch.nn.functional.conv2d(xr, wr, br, padding=1)), 2)
y.backward(d.float().cpu().permute(0, 3, 1, 2))
rel = lambda a, c: float((a - c).abs().max() / c.abs().max())
print(json.dumps({"dw_rel": rel(dw.cpu(), wr.grad), "db_rel": rel(db.cpu(), br.grad), "dw_norm": float(dw.norm()), "ref_norm": float(wr.grad.norm())}))
for _ in range(3): ops.conv1_bwd(x, w, b, d, dw, db)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n2 = 5120
x2 = x.repeat(10, 1, 1); d2 = d.repeat(10, 1, 1, 1)
for _ in range(2): ops.conv1_bwd(x2, w, b, d2, dw, db)
e0.record()
for _ in range(5): ops.conv1_bwd(x2, w, b, d2, dw, db)
e1.record(); torch.cuda.synchronize()
print("ms per call at 5120 clips:", e0.elapsed_time(e1) / 5)
dwa = dw.clone(); ops.conv1_bwd(x2, w, b, d2, dw, db); torch.cuda.synchronize(); print("deterministic:", bool(torch.equal(dwa, dw)))
