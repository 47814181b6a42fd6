import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
tb = importlib.import_module("train_bench")
print(tb.run(512, False, "bf16", steps=5))
