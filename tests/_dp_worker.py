"""Worker of the 2-rank data-parallel training test (tests/test_train_gpu.py): each rank runs
the fused HIP TrainStep on ITS half of every global batch; statistics and gradients are
exchanged with torch.distributed (gloo here, so that two ranks may share the single test GPU;
production uses the RCCL backend, same code path in ops.Dist)."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
PKG = "audio-classification-using-a-deep-cnn-combined-with-multi-level-attention_amd"


def main():
    out_path, steps, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    precision = sys.argv[4] if len(sys.argv) > 4 else "f32"
    finetune = len(sys.argv) > 5 and sys.argv[5] == "finetune"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mk = importlib.import_module("make_golden")
    W = importlib.import_module(PKG + ".weights")
    M = importlib.import_module(PKG + ".model")
    TR = importlib.import_module(PKG + ".train")
    ens = M.Ensemble("repeat", dict(mk.CNN_CONF), [2, 1], torch.device("cuda"), precision=precision)
    ens.load_state_dict({k: torch.as_tensor(v) for k, v in W.make_state_dict(7, W.ensemble_shapes((2, 1), False)).items()})
    ens.cuda()
    if finetune:
        M.set_requires_grad(ens, True)             # every parameter trainable: the gradient buckets travel while the CNN backward runs
    step = TR.TrainStep(ens, lr=1e-3)
    lo, hi = rank * B // world, (rank + 1) * B // world
    losses = []
    for s in range(steps):
        x, y = mk.synth_bags(100 + s, B)
        masks = mk.make_masks(200 + s, [2, 1], B)
        for lvl, em in enumerate(ens.mla.embedded_mappings):
            for j, d in enumerate(em.dropouts):
                d.mask = masks["mla.embedded_mappings.%d.dropouts.%d" % (lvl, j)][lo:hi]
        loss, hits = step(x[lo:hi].cuda(), y[lo:hi].cuda())
        losses.append(float(loss))
    sd = {k: v.detach().cpu().numpy() for k, v in ens.state_dict().items() if k.startswith("mla.") or (finetune and k.endswith(".bias"))}
    np.savez(out_path + ".rank%d.npz" % rank, losses=np.array(losses), **sd)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
