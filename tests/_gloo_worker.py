"""2-rank gloo worker (CPU): exercises ops.Dist (the exchange used by the data-parallel
training step and by SyncBN) and checks the data-parallel identities the HIP TrainStep relies
on, with the oracle as arithmetic: per-shard (sum, sum^2) all-reduced == global statistics, and
per-shard cross-entropy gradients scaled by 1/B_global all-reduced == global-batch gradient."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "audio-classification-using-a-deep-cnn-combined-with-multi-level-attention_amd"


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ops = importlib.import_module(PKG + ".ops")
    W = importlib.import_module(PKG + ".weights")
    d = ops.Dist()
    assert d.world == world
    B, T, Fd = 8, 10, 24
    x = torch.from_numpy(W.uniform(9, 1, B * T * Fd)).reshape(B, T, Fd).double()
    lo, hi = rank * B // world, (rank + 1) * B // world
    xs = x[lo:hi]
    sums = torch.stack([xs.sum(dim=(0, 2)), (xs * xs).sum(dim=(0, 2))], dim=1).reshape(-1).contiguous()
    d.all_reduce_sum(sums)
    full = torch.stack([x.sum(dim=(0, 2)), (x * x).sum(dim=(0, 2))], dim=1).reshape(-1)
    assert torch.allclose(sums, full, rtol=1e-12), "SyncBN sums"
    # loss scaling: local CE gradient with 1/B_global, summed over ranks == global mean-CE gradient
    scores = torch.from_numpy(W.uniform(9, 2, B * 10)).reshape(B, 10).double().requires_grad_(True)
    labels = torch.from_numpy(W.bits24(9, 3, B) % 10)
    F.cross_entropy(scores, labels).backward()
    ref = scores.grad.clone()
    local = torch.zeros_like(ref)
    s2 = scores.detach()[lo:hi].clone().requires_grad_(True)
    (F.cross_entropy(s2, labels[lo:hi], reduction="sum") / B).backward()
    local[lo:hi] = s2.grad
    d.all_reduce_sum(local)
    assert torch.allclose(local, ref, rtol=1e-12, atol=1e-15), "gradient all-reduce"
    dist.barrier()
    dist.destroy_process_group()
    print("rank %d ok" % rank)


if __name__ == "__main__":
    main()
