"""Eager vs HIP-graph training step (frozen bf16 CNN) over batch sizes: ms per step, same process.
    python scripts/train_graph_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
for bags in (8, 32, 128, 512):
    row = []
    for graph in (False, True):
        step, x, y = bench.make_train_step(bags, False, "bf16", 0, dev)
        step.use_graph = graph
        for _ in range(4):
            step(x, y)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 30
        for _ in range(n):
            step(x, y)
        torch.cuda.synchronize()
        row.append((time.perf_counter() - t0) / n * 1e3)
        del step
    print("%4d bags: eager %.3f ms, graph %.3f ms per step (%.2fx)" % (bags, row[0], row[1], row[0] / row[1]))
