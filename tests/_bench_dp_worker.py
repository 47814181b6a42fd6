"""One-rank NCCL (= RCCL) rehearsal of bench.dp_train_leg: the data-parallel train leg of `python bench.py --gpus N` with the
production transport (mla_allreduce_flat on a communicator from mla_comm_init_rank), the collectives forced on although the group has
one rank (MLA_DIST_ALWAYS=1): exercises the event tracing on the compute and communication streams, the bucketed exchange with its
exposed-wait events, ncclCommCount and the JSON sections -- everything but a second GPU."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    os.environ["MLA_DIST_ALWAYS"] = "1"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    import bench
    ts, coll = bench.dp_train_leg(1, 0, torch.device("cuda", 0), "nccl", bags=16, steps=2, warmup=1)
    print(json.dumps({"train_step": ts, "collective": coll}))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
