"""CPU tests of host-side logic added in round 2 (no GPU, no library calls): gradient-bucket assignment of the data-parallel
step, label validation, the resampling filter design against the oracle's restatement, the PMC traffic file bench.py reads, and
the shape of the C ABI header (every entry point documented with the reference interface it replaces)."""

import importlib
import json
import os
import re

import numpy as np
import pytest
import torch

from conftest import PKG, ROOT


def test_gradient_buckets_follow_the_backward_order():
    T = importlib.import_module(PKG + ".train")
    W = importlib.import_module(PKG + ".weights")
    names = list(W.ensemble_shapes((2, 1), False))
    params = [n for n in names if not any(s in n for s in ("running_", "num_batches"))]
    buckets = {}
    for n in params:
        buckets.setdefault(T.TrainStep._bucket_of(n), []).append(n)
    assert set(buckets) == {"mla", "conv14", "conv56", "fc0", "fc12"}
    assert all(n.startswith("mla.") for n in buckets["mla"])
    assert {n.split(".")[3] for n in buckets["conv14"]} == {"0", "3", "6", "8"} and {n.split(".")[3] for n in buckets["conv56"]} == {"11", "13"}
    assert {n.split(".")[3] for n in buckets["fc0"]} == {"0"} and {n.split(".")[3] for n in buckets["fc12"]} == {"2", "4"}
    # every CNN bucket has a trigger layer, and triggers fire in backward order (Linear 2 -> Linear 0 -> conv5 -> conv1)
    trig = T.TrainStep.BUCKET_TRIGGER
    assert set(trig) == set(buckets) - {"mla"}
    assert [b for b, _ in sorted(trig.items(), key=lambda kv: -kv[1])] == ["fc12", "fc0", "conv56", "conv14"]
    # state_dict order == forward order: each bucket is one contiguous run of parameters
    order = [T.TrainStep._bucket_of(n) for n in params]
    runs = [b for i, b in enumerate(order) if i == 0 or order[i - 1] != b]
    assert len(runs) == len(set(runs)) == 5


def test_label_validation_matches_cross_entropy_loss():
    ops = importlib.import_module(PKG + ".ops")
    ops.check_labels(torch.tensor([0, 9, 3]), 10)
    ops.check_labels(torch.empty(0, dtype=torch.long), 10)
    for bad in ([0, 10], [-1, 2]):
        with pytest.raises(IndexError):
            ops.check_labels(torch.tensor(bad), 10)
        with pytest.raises(IndexError):                                     # nn.CrossEntropyLoss raises the same type
            torch.nn.CrossEntropyLoss()(torch.zeros(len(bad), 10), torch.tensor(bad))
    with pytest.raises(IndexError):                                         # torch's ignore_index is refused, not honoured
        ops.check_labels(torch.tensor([-100]), 10)
    assert ops.raise_on_bad_labels(torch.tensor([7, 0])) == 7               # [n_correct, n_labels_out_of_range]
    with pytest.raises(IndexError):
        ops.raise_on_bad_labels(torch.tensor([7, 2]))
    # summed over data-parallel ranks the two counters cannot cancel: a rank with bad labels + a rank with hits still raises
    with pytest.raises(IndexError):
        ops.raise_on_bad_labels(torch.tensor([0, 1]) + torch.tensor([5, 0]))


def test_resampling_filter_is_the_restated_kaiser_best_design():
    fe = importlib.import_module(PKG + ".frontend")
    from oracle import resample as ors
    win, num_table = fe.kaiser_best_filter()
    ref, n_ref = ors.sinc_window()
    assert num_table == n_ref == 512 and np.array_equal(win, ref) and win.dtype == np.float64


def test_pmc_traffic_file_feeds_the_bench_line():
    import glob
    path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))[-1]     # bench.py reads the newest round's file
    d = json.load(open(path))
    for key in ("conv2/bf16", "conv3/bf16", "conv4/bf16", "conv5/bf16", "conv6/bf16", "logmel/bf16"):
        assert d[key]["bytes_per_clip"] > 0 and "FETCH_SIZE" in d[key]["source"] and "WRITE_SIZE" in d[key]["source"]
    # measured traffic is never below the algorithmic activation bytes (bench.py CONV_BYTES / FE_BYTES); above them by the
    # weight re-fetches of the MFMA-bound layers (conv6's 4.7 MB of weights do not stay in a 4 MB L2), within 3x
    algo = {"conv2/bf16": (48 * 32 * 64 + 24 * 16 * 128) * 2, "conv3/bf16": (24 * 16 * 128 + 24 * 16 * 256) * 2,
            "conv4/bf16": (24 * 16 * 256 + 12 * 8 * 256) * 2, "conv5/bf16": (12 * 8 * 256 + 12 * 8 * 512) * 2,
            "conv6/bf16": (12 * 8 * 512 + 6 * 4 * 512) * 2, "logmel/bf16": 15360 * 4 + 96 * 64 * 2}
    for key, a in algo.items():
        assert a <= d[key]["bytes_per_clip"] <= 3 * a, (key, d[key]["bytes_per_clip"], a)


def test_every_abi_entry_point_cites_what_it_replaces():
    text = open(os.path.join(ROOT, "include", "mla_hip.h")).read()
    lib = importlib.import_module(PKG + "._lib")
    syms = lib.declared_symbols()
    assert len(syms) >= 60
    # the header cites reference files (file:line) throughout, and the integration table lists every symbol family
    assert len(re.findall(r"\b(?:mel_features|vggish_input|vggish|model|train|dataset)\.py:\d+", text)) >= 40
    integ = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    def listed(sym):                 # by name, or as a `(_suffix)` variant of a listed family: mla_bn_stats(_sums/_finish)
        parts = sym.split("_")
        return any("_".join(parts[:k]) in integ for k in range(len(parts), 1, -1) if k > 2 or k == len(parts))
    missing = [s for s in syms if not listed(s) and not s.startswith(("mla_abi", "mla_last", "mla_logmel_table", "mla_logmel_reference"))]
    assert not missing, missing
