"""world_size-2 gloo test on CPU of the data-parallel exchange layer (ops.Dist) and of the
identities the multi-GPU training step is built on (see tests/_gloo_worker.py). The same
TrainStep code path runs under RCCL on the GPUs; tests/test_train_gpu.py runs it with two ranks
on a real GPU."""

import os
import subprocess
import sys

from conftest import ROOT


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_gloo_exchange():
    env = dict(os.environ, WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_gloo_worker.py")],
                              env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    for r, p in enumerate(procs):
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0, out.decode()
        assert ("rank %d ok" % r) in out.decode()
