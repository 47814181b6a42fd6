"""GPU test of the bench contract: `python bench.py` at a reduced batch prints ONE JSON line with the fields the driver parses
(metric / value / unit / n_gpus / steps / warmup / ms_per_step / scaling / dtype / config.workload, `roofline` and `cpu_baseline`),
with internally consistent numbers. The full-size numbers live in profiles/; this checks the plumbing, including the self-launched
two-rank path (gloo rehearsal on one GPU: the N > 1 branch of the timing code)."""

import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def run_bench(*args):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=env, cwd=ROOT, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_one_gpu_line_has_the_contract_fields_and_consistent_numbers():
    d = run_bench("--bags", "64", "--steps", "3", "--warmup", "1", "--prewarm-seconds", "0.2", "--no-small-batch", "--no-h2d", "--no-train-leg")
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["unit"] == "clips/s" and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["dtype"] == "bf16" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    clips = d["config"]["clips_per_step_per_gpu"]
    assert clips == 640 and abs(d["value"] - clips / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - r["flop_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e12) < 1e-6 * r["achieved"]
    assert r["traffic"] is not None and r["traffic"] >= r["algorithmic_bytes"] > 0          # PMC bytes per clip x clips of this launch
    assert sum(d["kernel_ms"].values()) <= d["ms_per_step"] * 1.02                              # per-kernel events inside the timed region
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert c["parity_max_rel_f32"] < 1e-4 and c["parity_max_rel_bf16x3"] < 1e-4 and c["parity_max_rel_bf16"] < 1e-2
    v = d["value_at_tolerance"]
    assert v["mode"] in ("bf16x3", "f32") and v["measured_max_rel"] <= v["tolerance"] == 1e-4
    assert d["roofline_frontend"]["bound"] == "hbm" and 0 < d["roofline_frontend"]["frac"] < 1
    # the bound that binds the front-end (vector issue, LDS array), from the PMC pass on file x the clips of this launch
    i = d["roofline_frontend"]["issue"]
    assert 10000 < i["valu_insts_per_clip"] < 25000 and 0 < i["frac_of_valu_bound"] < 1 and 0 < i["frac_of_lds_bound"] < 1
    assert abs(i["valu_bound_ms"] - i["valu_insts_per_clip"] * clips * 2 / (1024 * 2.4e9) * 1e3) < 1e-9


def test_two_self_launched_ranks_report_the_aggregate_and_the_data_parallel_train_step():
    """`python bench.py --gpus 2` (the driver's scaling invocation; gloo here, two ranks sharing the test GPU): the inference
    headline AND BASELINE config 5's data-parallel train step with the evidence of its collectives."""
    d = run_bench("--gpus", "2", "--backend", "gloo", "--bags", "32", "--steps", "2", "--warmup", "1", "--prewarm-seconds", "0.2",
                  "--train-bags", "32", "--train-steps", "2")
    assert d["n_gpus"] == 2 and d["config"]["clips_per_step_per_gpu"] == 320
    assert abs(d["value"] - 2 * 320 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]       # whole-job rate over max-over-ranks time
    assert "cpu_baseline" not in d
    ts, co = d["train_step"], d["collective"]
    assert ts["bags_per_gpu"] == 32 and ts["global_batch_bags"] == 64 and ts["clips_per_step"] == 640 and ts["n_gpus"] == 2
    for name in ("frozen_bf16", "frozen_bf16_per_shard_bn", "finetune_bf16"):
        v, c = ts[name], co[name]
        assert v["ms_per_step"] > 0 and abs(v["clips_per_s"] - 640 / (v["ms_per_step"] * 1e-3)) < 1e-6 * v["clips_per_s"]
        assert 0 < v["frac"] < 1 and abs(v["loss_last"] - 2.3) < 0.3 and 0 <= v["hits_last_global"] <= 64
        assert c["ranks"] == 2 and c["active"] and c["backend"] == "gloo" and "host copies" in c["transport"]
        assert c["allreduce_bytes_per_step"] >= 4 * v["trainable_floats"] and c["allreduce_ms"] > 0 and c["bus_GBps"] > 0
        assert c["other_allreduces_per_step"] == 2                                            # loss + hit counters
        assert 0.0 <= c["overlap_hidden_frac"] <= 1.0
    assert ts["frozen_bf16"]["trainable_floats"] == 823050 - 2 * 6010 and ts["finetune_bf16"]["trainable_floats"] == 72964234 - 2 * 6010
    # SyncBN: 9 forward statistics (normv / normf share theirs) + 10 backward sums per step for MLA [2, 1]; none per shard
    assert co["frozen_bf16"]["syncbn_allreduces_per_step"] == co["finetune_bf16"]["syncbn_allreduces_per_step"] > 10
    assert co["frozen_bf16_per_shard_bn"]["syncbn_allreduces_per_step"] == 0 and not co["frozen_bf16_per_shard_bn"]["sync_bn"]
    assert set(co["finetune_bf16"]["gradient_messages_per_step"]) == {"mla", "fc12", "fc0", "conv56", "conv14"}
    assert set(co["frozen_bf16"]["gradient_messages_per_step"]) == {"flat"}
    assert "second HIP stream" in co["finetune_bf16"]["gradient_exchange"]


def test_train_mode_two_ranks_reports_the_collectives():
    d = run_bench("--gpus", "2", "--backend", "gloo", "--mode", "train", "--bags", "16", "--steps", "2", "--warmup", "1")
    assert d["n_gpus"] == 2 and d["config"]["global_batch_bags"] == 32 and d["collective"]["ranks"] == 2
    assert d["collective"]["syncbn_allreduces_per_step"] > 10 and d["collective"]["allreduce_bytes_per_step"] >= 4 * 811030


def test_data_parallel_train_leg_over_rccl_on_one_rank():
    """bench.dp_train_leg with backend nccl on a one-rank group, collectives forced on (tests/_bench_dp_worker.py): the production
    transport end to end inside the bench leg -- what a box with one GPU can execute of BASELINE config 5."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONDONTWRITEBYTECODE="1")
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        env["MASTER_PORT"] = str(so.getsockname()[1])
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_bench_dp_worker.py")], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    d = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1])
    ts, co = d["train_step"], d["collective"]
    for name in ("frozen_bf16", "frozen_bf16_per_shard_bn", "finetune_bf16"):
        c = co[name]
        assert c["ranks"] == 1 and c["backend"] == "nccl" and "mla_allreduce_flat" in c["transport"] and c["fallback"] is None
        assert c["rccl_library_origin"] == "already loaded by the host"
        assert c["allreduce_bytes_per_step"] >= 4 * ts[name]["trainable_floats"] and c["allreduce_ms"] > 0
        assert c["other_allreduces_per_step"] == 2 and abs(ts[name]["loss_last"] - 2.3) < 0.3
    assert co["frozen_bf16"]["syncbn_allreduces_per_step"] == 18 and co["frozen_bf16_per_shard_bn"]["syncbn_allreduces_per_step"] == 0
    f = co["finetune_bf16"]
    assert set(f["gradient_messages_per_step"]) == {"mla", "fc12", "fc0", "conv56", "conv14"} and 0.0 <= f["overlap_hidden_frac"] <= 1.0
    assert f["gradient_messages_per_step"]["fc0"]["bytes"] == 4 * (12288 * 4096 + 4096)
