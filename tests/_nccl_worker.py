"""One-rank RCCL worker (tests/test_train_gpu.py::test_rccl_branch_...): backend "nccl" on the test GPU, world size 1, with
the collectives forced on (ops.Dist(always=True) / MLA_DIST_ALWAYS=1), so that the production transport of the data-parallel
step -- mla_allreduce_flat on a communicator made by mla_comm_init_rank, and torch.distributed's all_reduce on the device
buffer as the alternative -- really executes: float32 / float64 / int32 buffers, and the whole TrainStep (SyncBN sums,
flat gradient, loss, hit count). A one-rank sum must leave every value unchanged, bit for bit."""
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


def run_steps(mk, W, M, TR, always, finetune):
    os.environ["MLA_DIST_ALWAYS"] = "1" if always else "0"
    ens = M.Ensemble("repeat", dict(mk.CNN_CONF), [2, 1], torch.device("cuda"))
    ens.load_state_dict({k: torch.as_tensor(v) for k, v in W.make_state_dict(7, W.ensemble_shapes((2, 1), False)).items()})
    ens.cuda()
    if finetune:
        M.set_requires_grad(ens, True)
    step = TR.TrainStep(ens, lr=1e-3)
    assert step.dist.active == always
    if finetune:                                   # the five gradient buckets tile the flat buffer, in forward order
        spans = sorted(step.buckets.values())
        assert set(step.buckets) == {"mla", "conv14", "conv56", "fc0", "fc12"} and spans[0][0] == 0 and spans[-1][1] == step.flat_g.numel()
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert [k for k, _ in sorted(step.buckets.items(), key=lambda kv: kv[1])] == ["mla", "conv14", "conv56", "fc0", "fc12"]
    losses, hits = [], []
    for s in range(3):
        x, y = mk.synth_bags(100 + s, 4)
        masks = mk.make_masks(200 + s, [2, 1], 4)
        for lvl, em in enumerate(ens.mla.embedded_mappings):
            for j, d in enumerate(em.dropouts):
                d.mask = masks["mla.embedded_mappings.%d.dropouts.%d" % (lvl, j)]
        loss, h = step(x.cuda(), y.cuda())
        losses.append(float(loss)); hits.append(h.tolist())
    via = step.dist.via
    step.dist.close()
    return losses, hits, step.flat_p.clone(), via


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    ops = importlib.import_module(PKG + ".ops")
    L = importlib.import_module(PKG + "._lib")
    mk = importlib.import_module("make_golden")
    W = importlib.import_module(PKG + ".weights")
    M = importlib.import_module(PKG + ".model")
    TR = importlib.import_module(PKG + ".train")
    for via in ("abi", "torch"):
        os.environ["MLA_DIST_COLLECTIVE"] = via
        d = ops.Dist(always=True)
        assert d.active and d.world == 1 and d.via == via, (d.active, d.world, d.via)
        if via == "abi":
            origin = L.lib().mla_comm_library_origin().decode()
            assert origin == "already loaded by the host", origin        # PyTorch's RCCL, not a second one
            assert d.comm is not None
        for t in (torch.from_numpy(W.uniform(3, 1, 100003)).cuda(), torch.from_numpy(W.uniform(3, 2, 40, dtype=np.float64)).cuda(),
                  torch.arange(-5, 6, dtype=torch.int32).cuda()):
            want = t.clone()
            got = d.all_reduce_sum(t)
            torch.cuda.synchronize()
            assert got is t and torch.equal(t, want), (via, t.dtype)
        d.close()
        base = run_steps(mk, W, M, TR, False, False)
        coll = run_steps(mk, W, M, TR, True, False)
        assert base[3] is None and coll[3] == via
        assert base[0] == coll[0] and base[1] == coll[1] and torch.equal(base[2], coll[2]), via
        print("nccl worker: %s transport ok, losses %s" % (via, coll[0]))
    os.environ["MLA_DIST_COLLECTIVE"] = "abi"
    # a rank that cannot create its communicator through the C ABI: the ranks agree (MIN all-reduce) and ALL use torch's transport

    d = ops.Dist(always=True)
    assert d.via == "abi" and d.fallback is None and d.ranks_reported() == 1 and d.describe()["ranks"] == 1
    d.close()
    orig_init = ops.Dist._init_comm
    ops.Dist._init_comm = lambda self, dist_mod: (_ for _ in ()).throw(RuntimeError("simulated: no ncclCommInitRank in this RCCL"))
    try:
        d = ops.Dist(always=True)
    finally:
        ops.Dist._init_comm = orig_init
    assert d.via == "torch" and d.comm is None and "simulated" in d.fallback and d.describe()["fallback"] == d.fallback
    t = torch.arange(7, dtype=torch.float32).cuda()
    assert torch.equal(d.all_reduce_sum(t.clone()), t)
    print("nccl worker: agreed fallback to torch's transport ok (%s)" % d.fallback)
    base = run_steps(mk, W, M, TR, False, True)
    for overlap in ("1", "0"):                     # bucketed reductions on the communication stream / one flat all-reduce at the end
        os.environ["MLA_DIST_OVERLAP"] = overlap
        coll = run_steps(mk, W, M, TR, True, True)
        assert base[0] == coll[0] and torch.equal(base[2], coll[2]), overlap
    print("nccl worker: finetune step over the C-ABI all-reduce ok (bucketed + overlapped, and flat)")
    dist.barrier()
    dist.destroy_process_group()
    print("nccl worker ok")


if __name__ == "__main__":
    main()
