"""Target of rocprofv3 --kernel-trace --stats for the training step: 3 warm-up + 5 profiled steps of the finetune (default bf16) or frozen step.
    python scripts/train_prof.py [bf16|f32] [finetune|frozen]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
finetune = (sys.argv[2] if len(sys.argv) > 2 else "finetune") == "finetune"
step, x, y = bench.make_train_step(512, finetune, prec, 0, torch.device("cuda", 0))
for _ in range(8):
    loss, _ = step(x, y)
torch.cuda.synchronize()
print(float(loss))
