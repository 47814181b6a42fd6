"""`python bench.py --gpus N` from a plain shell must launch its own ranks (the driver's scaling run starts it that way):
N child processes with the torch.distributed.run environment, spawned before the parent makes any GPU call, rank 0's
JSON line relayed, a failing rank's exit code propagated and the surviving ranks stopped. `--dry-run` replaces the GPU
work by a gloo rendezvous + one all-reduce, so the launch path itself runs here on CPU."""

import json
import os
import subprocess
import sys

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _run(*args, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run([sys.executable, BENCH] + list(args), env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    return p.returncode, lines, p.stderr.decode()


def test_self_launch_two_ranks_prints_one_line():
    rc, lines, err = _run("--gpus", "2", "--dry-run", "--backend", "gloo")
    assert rc == 0, err
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rank_sum"] == 1.0          # ranks 0 + 1 took part in the all-reduce


def test_default_line_at_n_gt_1_carries_the_data_parallel_train_step():
    """The driver's scaling run is `python bench.py --gpus N` with no mode flag: at N > 1 that line must hold the inference headline
    AND the data-parallel train step (BASELINE config 5) with its collective section; flags reach the ranks."""
    rc, lines, err = _run("--gpus", "2", "--dry-run", "--backend", "gloo")
    assert rc == 0, err
    d = json.loads(lines[0])
    assert d["legs"] == ["headline", "train_step", "collective"] and d["train_bags"] == 512
    assert d["dp_variants"] == ["frozen_bf16", "frozen_bf16_per_shard_bn", "finetune_bf16"]
    rc, lines, err = _run("--gpus", "2", "--dry-run", "--backend", "gloo", "--train-bags", "32", "--train-steps", "3", "--train-leg-timeout", "99")
    d = json.loads(lines[0])
    assert (d["train_bags"], d["train_steps"], d["train_leg_timeout"]) == (32, 3, 99)
    rc, lines, err = _run("--gpus", "2", "--dry-run", "--backend", "gloo", "--no-train-leg")
    assert json.loads(lines[0])["legs"] == ["headline"]
    rc, lines, err = _run("--dry-run")
    assert json.loads(lines[0])["legs"] == ["headline", "train_step"]


def test_watchdog_prints_what_it_has_and_leaves(tmp_path):
    """bench.Watchdog: a leg stuck in a collective never returns to Python; the timer thread lets rank 0 print its line and exits 0."""
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "with bench.Watchdog(0.5, lambda: print('{\"fired\": true}', flush=True)):\n"
            "    time.sleep(30)\n"
            "print('not reached')\n") % ROOT
    p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode == 0 and p.stdout.decode().strip() == '{"fired": true}'
    code = ("import sys; sys.path.insert(0, %r); import bench\n"
            "with bench.Watchdog(30, lambda: print('fired')):\n"
            "    pass\n"
            "print('done')\n") % ROOT
    p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode == 0 and p.stdout.decode().strip() == "done"


def test_self_launch_train_mode_and_single():
    rc, lines, err = _run("--gpus", "2", "--dry-run", "--mode", "train")
    assert rc == 0 and json.loads(lines[0])["mode"] == "train", err
    rc, lines, err = _run("--dry-run")
    assert rc == 0 and json.loads(lines[0])["n_gpus"] == 1, err


def test_failing_rank_propagates_and_stops_the_others():
    rc, lines, err = _run("--gpus", "3", "--dry-run", "--dry-run-fail-rank", "1")
    assert rc == 7 and "rank 1 exited with code 7" in err
    assert lines == []


def test_under_torchrun_environment_no_respawn():
    """Started as ONE of the ranks (WORLD_SIZE set, as torch.distributed.run does): no children; a mismatch with --gpus is an error."""
    rc, lines, err = _run("--gpus", "1", "--dry-run", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc == 0 and len(lines) == 1, err
    rc, lines, err = _run("--gpus", "2", "--dry-run", env={"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc != 0 and "WORLD_SIZE=4" in err


def test_parent_makes_no_gpu_call_before_spawning():
    """Static check of the order in main(): launch_ranks() is reached before any torch.cuda call."""
    src = open(BENCH).read()
    main = src[src.index("def main():"):src.index("def load_pmc_traffic")]
    assert main.index("launch_ranks(") < main.index("torch.cuda.")
    launch = src[src.index("def launch_ranks("):src.index("def dry_run(")]
    assert "torch.cuda" not in launch and "os.exec" not in launch


def test_collective_summary_arithmetic():
    """bench.collective_summary on a synthetic trace (no GPU): per-step bytes / ms, bus GB/s = 2 (N - 1) / N x bytes / time, SyncBN
    counts, overlap_hidden_frac = 1 - exposed wait / all-reduce time."""
    import sys
    sys.path.insert(0, ROOT)
    import bench

    class Ev:
        def __init__(self, t):
            self.t = t

        def elapsed_time(self, other):
            return other.t - self.t

    class FakeDist:
        trace = []

        def describe(self):
            return {"ranks": 8, "transport": "x"}

    class FakeStep:
        dist, exposed = FakeDist(), []

    steps, world = 2, 8
    for s in range(steps):
        for tag, nbytes, ms in (("grad:mla", 3_244_304, 0.1), ("grad:fc0", 201_342_976, 2.0), ("syncbn_fwd", 160, 0.02), ("syncbn_bwd", 160, 0.03),
                                ("loss", 4, 0.01), ("hits", 8, 0.01)):
            FakeStep.dist.trace.append((tag, nbytes, Ev(0.0), Ev(ms)))
        FakeStep.exposed.append((Ev(0.0), Ev(0.42)))
    c = bench.collective_summary(FakeStep, steps, world)
    assert c["ranks"] == 8 and c["allreduce_bytes_per_step"] == 3_244_304 + 201_342_976 and abs(c["allreduce_ms"] - 2.1) < 1e-12
    assert abs(c["bus_GBps"] - 2 * 7 / 8 * (3_244_304 + 201_342_976) / 2.1e-3 / 1e9) < 1e-9
    assert c["syncbn_allreduces_per_step"] == 2 and abs(c["syncbn_allreduce_ms"] - 0.05) < 1e-12 and c["syncbn_bytes_per_allreduce"] == 160
    assert c["other_allreduces_per_step"] == 2 and set(c["gradient_messages_per_step"]) == {"mla", "fc0"}
    assert abs(c["exposed_wait_ms"] - 0.42) < 1e-12 and abs(c["overlap_hidden_frac"] - (1 - 0.42 / 2.1)) < 1e-12
