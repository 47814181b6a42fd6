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


def test_two_self_launched_ranks_report_the_aggregate():
    d = run_bench("--gpus", "2", "--backend", "gloo", "--bags", "32", "--steps", "2", "--warmup", "1", "--prewarm-seconds", "0.2")
    assert d["n_gpus"] == 2 and d["config"]["clips_per_step_per_gpu"] == 320
    assert abs(d["value"] - 2 * 320 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]       # whole-job rate over max-over-ranks time
    assert "cpu_baseline" not in d and "train_step" not in d                                  # N > 1: the headline line only
