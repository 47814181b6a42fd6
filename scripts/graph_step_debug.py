"""Debug: per-step checksums of eager vs eager vs graph training steps (frozen bf16, no injected masks)."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden as mk
PKG = "audio-classification-using-a-deep-cnn-combined-with-multi-level-attention_amd"
W = importlib.import_module(PKG + ".weights")
from test_train_graph_gpu import make

def trace(graph, ords, steps=6, B=16):
    ens, step, ords = make(mk, W, "bf16", False, graph, ords)
    rows = []
    for s in range(steps):
        x, y = mk.synth_bags(300 + s, B)
        loss, hits = step(x.cuda(), y.cuda())
        rows.append((float(loss), float(step.flat_g.double().abs().sum()), float(step.flat_p.double().sum()), float(step.last_out.double().sum()),
                     float(ens.mla.norm.running_mean.double().sum())))
    return rows, ords
a, ords = trace(False, None)
b, _ = trace(False, ords)
c, _ = trace(True, ords)
for i, (x, y, z) in enumerate(zip(a, b, c)):
    print(i, "eager==eager", x == y, "eager==graph", x == z)
    if x != z:
        print("   eager", x); print("   graph", z)
