"""The training step as one HIP graph (TrainStep(graph=True), the default without a process group) against the same steps run
eagerly: Adam's step-dependent scalars and the dropout call number are read from device memory inside the graph (mla_adam_step_dev,
mla_dropout_mask_dev), so replays must reproduce the eager sequence bit for bit -- losses, hit counts, every updated parameter,
running statistics, the masks drawn -- also across a change of batch size (re-capture), a checkpoint restore and an eager step
in between."""

import importlib

import numpy as np
import pytest
import torch

from conftest import PKG

pytestmark = pytest.mark.gpu


def make(mk, W, precision, finetune, graph, ordinals=None):
    M = importlib.import_module(PKG + ".model")
    TR = importlib.import_module(PKG + ".train")
    torch.manual_seed(77)                                        # Dropout reads the seed at construction
    ens = M.Ensemble("repeat", dict(mk.CNN_CONF), [2, 1], torch.device("cuda"), precision=precision)
    ens.load_state_dict({k: torch.as_tensor(v) for k, v in W.make_state_dict(7, W.ensemble_shapes((2, 1), False)).items()})
    ens.cuda()
    drops = [m for m in ens.mla.modules() if type(m).__name__ == "Dropout"]
    if ordinals is not None:                                     # the mask stream is keyed by the module's ordinal: same for both runs
        for d, o in zip(drops, ordinals):
            d.ordinal = o
    if finetune:
        M.set_requires_grad(ens, True)
    return ens, TR.TrainStep(ens, lr=1e-3, graph=graph), [d.ordinal for d in drops]


def run(step, mk, plan):
    out = []
    for s, B in plan:
        x, y = mk.synth_bags(300 + s, B)
        loss, hits = step(x.cuda(), y.cuda() if s % 2 else y)     # device and host labels alternate
        out.append((float(loss), hits.tolist()))
    return out


@pytest.mark.parametrize("precision,finetune", [("bf16", False), ("f32", False), ("bf16", True)])
def test_graph_replays_equal_eager_steps(mk, W, precision, finetune):
    plan = [(0, 16), (1, 16), (2, 16), (3, 16), (4, 8), (5, 8), (6, 8), (7, 16), (8, 16)]        # (step, bags): two re-captures
    ens_e, step_e, ords = make(mk, W, precision, finetune, graph=False)
    ref = run(step_e, mk, plan)
    assert step_e._graph is None
    ens_g, step_g, _ = make(mk, W, precision, finetune, graph=True, ordinals=ords)
    got = run(step_g, mk, plan)
    assert step_g._graph is not None and step_g._graph["shape"][0] == 16 and step_g.t == len(plan)
    assert got == ref, (got, ref)
    assert torch.equal(step_g.flat_p, step_e.flat_p) and torch.equal(step_g.flat_m, step_e.flat_m) and torch.equal(step_g.flat_v, step_e.flat_v)
    for (k, a), (_, b) in zip(ens_g.state_dict().items(), ens_e.state_dict().items()):
        assert torch.equal(a, b), k                               # running statistics and num_batches_tracked included
    assert int(step_g.step_dev) == len(plan)
    # eval-mode forward after graph steps sees the updated weights (derived copies are refreshed), same as after eager steps
    ens_g.eval(); ens_e.eval()
    x = mk.synth_bags(999, 4)[0].cuda()
    with torch.no_grad():
        assert torch.equal(ens_g(x), ens_e(x))


def test_graph_survives_restore_and_foreign_mask_draws(mk, W):
    """load_state_dict moves the step count; a train-mode forward outside the step advances the dropout call numbers: the next
    step must notice both (device counter refreshed, graph re-captured) and continue like the eager twin."""
    ens_e, step_e, ords = make(mk, W, "bf16", False, graph=False)
    ens_g, step_g, _ = make(mk, W, "bf16", False, graph=True, ordinals=ords)
    plan = [(0, 8), (1, 8), (2, 8)]
    assert run(step_g, mk, plan) == run(step_e, mk, plan)
    sd = step_e.state_dict()
    sd["step"] = torch.tensor(40)
    step_e.load_state_dict(sd); step_g.load_state_dict({k: v.clone() for k, v in sd.items()})
    x = mk.synth_bags(5, 8)[0].cuda()
    with torch.no_grad():
        for ens in (ens_e, ens_g):
            ens.train()
            ens(x)                                                # draws masks, bumps running statistics -- on both twins alike
    plan = [(3, 8), (4, 8)]
    assert run(step_g, mk, plan) == run(step_e, mk, plan)
    assert step_g.t == step_e.t == 42 and torch.equal(step_g.flat_p, step_e.flat_p)


def test_device_counter_kernels_match_the_host_forms(W):
    """mla_dropout_mask_dev == mla_dropout_mask at stream_base + counter + 1; mla_adam_prepare + mla_adam_step_dev == mla_adam_step
    (bit for bit over 30 steps; the call counter advances by itself)."""
    ops = importlib.import_module(PKG + ".ops")
    dev = torch.device("cuda")
    ctr = torch.tensor([6], dtype=torch.int64, device=dev)
    for n, seed, base, p in ((4097, 123456789, (3 << 32) + 11, 0.4), (65, 2 ** 63 + 5, 0, 0.5)):
        assert torch.equal(ops.dropout_mask_dev(n, seed, base, ctr, 17, p, dev), ops.dropout_mask(n, seed, base + 7, 17, p, dev))
    n = 10007
    g = torch.from_numpy(W.uniform(3, 1, n, lo=-1e-2, hi=1e-2)).to(dev)
    state = [[torch.from_numpy(W.uniform(3, 2, n)).to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)] for _ in range(2)]
    ctr = torch.zeros(1, dtype=torch.int64, device=dev)
    scal = torch.zeros(2, device=dev)
    for t in range(1, 31):
        ops.adam_step(state[0][0], g, state[0][1], state[0][2], 1e-3, 0.9, 0.999, 1e-8, t)
        ops.adam_prepare(scal, 1e-3, 0.9, 0.999, t)
        ops.adam_step_dev(state[1][0], g, state[1][1], state[1][2], 0.9, 0.999, 1e-8, scal, ctr)
        g = g * 0.97 + 1e-4
    assert int(ctr) == 30
    for a, b in zip(state[0], state[1]):
        assert torch.equal(a, b)
