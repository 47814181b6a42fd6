"""GPU tests of the fused HIP training step (train.py:119-142 semantics) against the loss curve,
gradients and updated parameters that the REFERENCE produced (tests/golden/train.npz), and of
the data-parallel path (2 ranks == 1 rank on the global batch)."""

import importlib
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import PKG, ROOT

pytestmark = pytest.mark.gpu

NOISY = ("fc.bias", "fc.0.bias", "fc.1.bias", "fcv.bias")     # zero-gradient biases: see test_oracle_golden.py


def build(mk, W):
    M = importlib.import_module(PKG + ".model")
    ens = M.Ensemble("repeat", dict(mk.CNN_CONF), [2, 1], torch.device("cuda"))
    ens.load_state_dict({k: torch.as_tensor(v) for k, v in W.make_state_dict(7, W.ensemble_shapes((2, 1), False)).items()})
    return ens.cuda()


def install(ens, masks, lo=None, hi=None):
    for lvl, em in enumerate(ens.mla.embedded_mappings):
        for j, d in enumerate(em.dropouts):
            d.mask = masks["mla.embedded_mappings.%d.dropouts.%d" % (lvl, j)][lo:hi]


def test_training_curve_matches_reference_golden(golden, mk, W):
    g = golden("train")
    TR = importlib.import_module(PKG + ".train")
    ens = build(mk, W)
    step = TR.TrainStep(ens, lr=1e-3)
    assert step.n_params == 823050 - 2 * 6010, "trainable minus the dead fcf parameters"
    losses = []
    for s in range(10):
        x, y = mk.synth_bags(100 + s, 8)
        install(ens, mk.make_masks(200 + s, [2, 1], 8))
        loss, hits = step(x.cuda(), y.cuda())
        losses.append(float(loss))
        if s == 0:
            np.testing.assert_allclose(step.last_out.cpu().numpy(), g["frozen/out_first"], rtol=1e-4, atol=1e-5)
            # (grads are read after Adam ran; Adam does not modify them)
            for name, gr in step.grads.items():
                ref = float(g["frozen/gradnorm0/" + name])
                got = float(gr.double().norm())
                if ref < 1e-4:
                    assert got < 1e-4, name
                else:
                    assert got == pytest.approx(ref, rel=2e-3), name
            np.testing.assert_allclose(step.grads["mla.fc.weight"].cpu().numpy(), g["frozen/grad0/mla.fc.weight"], rtol=1e-3, atol=1e-6)
            np.testing.assert_allclose(step.grads["mla.embedded_mappings.0.norm0.weight"].cpu().numpy(),
                                       g["frozen/grad0/mla.embedded_mappings.0.norm0.weight"], rtol=1e-3, atol=1e-6)
            np.testing.assert_allclose(step.grads["mla.attention_modules.1.fcv.weight"].cpu().numpy(),
                                       g["frozen/grad0/mla.attention_modules.1.fcv.weight"], rtol=1e-3, atol=1e-6)
    # the first steps pin the arithmetic; later ones only bound the drift: Adam turns last-bit differences
    # (summation order of a reduction) of near-zero gradients into +-lr steps, see test_oracle_golden.py
    np.testing.assert_allclose(losses[:3], g["frozen/losses"][:3], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(losses, g["frozen/losses"], rtol=2e-3, atol=1e-5)
    np.testing.assert_allclose(step.last_out.cpu().numpy(), g["frozen/out_last"], rtol=0, atol=2e-2)   # 10 Adam steps amplify last-bit differences ~3x per step
    sd = ens.state_dict()
    for k in g.files:
        if k.startswith("frozen/final/") and not k.endswith(NOISY):
            atol = 1e-2 if k.endswith("running_mean") else 2e-4
            np.testing.assert_allclose(sd[k[len("frozen/final/"):]].cpu().numpy(), g[k], rtol=1e-3, atol=atol, err_msg=k)
    fcf = sd["mla.attention_modules.0.fcf.bias"].cpu().numpy()
    assert np.array_equal(fcf, W.make_tensor(7, "mla.attention_modules.0.fcf.bias", (10,))), "dead parameters stay untouched"
    ens.eval()
    ev = ens(mk.synth_bags(999, 4)[0].cuda())
    np.testing.assert_allclose(ev.cpu().numpy(), g["frozen/eval_after"], rtol=0, atol=2e-2)


def test_training_step_is_deterministic(mk, W):
    """Every reduction has a fixed order (no float atomics): two runs give identical bits."""
    TR = importlib.import_module(PKG + ".train")
    runs = []
    for _ in range(2):
        ens = build(mk, W)
        step = TR.TrainStep(ens, lr=1e-3)
        losses = []
        for s in range(4):
            x, y = mk.synth_bags(100 + s, 8)
            install(ens, mk.make_masks(200 + s, [2, 1], 8))
            losses.append(float(step(x.cuda(), y.cuda())[0]))
        runs.append((losses, step.flat_p.clone()))
    assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1])


def test_two_ranks_equal_one_rank_on_the_global_batch(tmp_path, golden):
    """Data-parallel correctness by construction: SyncBN sums + 1/B_global loss scaling + one
    gradient all-reduce reproduce the single-process step. Two ranks share the test GPU (gloo)."""
    g = golden("train")
    out = str(tmp_path / "dp")
    env = dict(os.environ, WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dp_worker.py"), out, "5", "8"],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    r0, r1 = np.load(out + ".rank0.npz"), np.load(out + ".rank1.npz")
    np.testing.assert_allclose(r0["losses"][:3], g["frozen/losses"][:3], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(r0["losses"], g["frozen/losses"][:5], rtol=2e-3, atol=1e-5)
    np.testing.assert_array_equal(r0["losses"], r1["losses"])
    for k in r0.files:
        if k != "losses":
            np.testing.assert_array_equal(r0[k], r1[k], err_msg=k)      # replicas stay bit-identical


def test_finetune_curve_matches_reference_golden(golden, mk, W):
    """Finetune (train.py:96-97: every parameter trainable): CNN gradients from the HIP dgrad / wgrad /
    pool-backward kernels. 4 steps, 4 bags, against the reference's own run."""
    g = golden("train")
    TR = importlib.import_module(PKG + ".train")
    M = importlib.import_module(PKG + ".model")
    ens = build(mk, W)
    M.set_requires_grad(ens, True)
    step = TR.TrainStep(ens, lr=1e-3)
    assert step.n_params == 72964234 - 2 * 6010
    losses = []
    for s in range(4):
        x, y = mk.synth_bags(100 + s, 4)
        install(ens, mk.make_masks(200 + s, [2, 1], 4))
        loss, hits = step(x.cuda(), y.cuda())
        losses.append(float(loss))
        if s == 0:
            np.testing.assert_allclose(step.last_out.cpu().numpy(), g["finetune/out_first"], rtol=1e-4, atol=1e-5)
            worst = 0.0
            for name, gr in step.grads.items():
                ref = float(g["finetune/gradnorm0/" + name])
                got = float(gr.double().norm())
                if ref < 1e-4:
                    assert got < 1e-4, name
                else:
                    worst = max(worst, abs(got - ref) / ref)
                    assert got == pytest.approx(ref, rel=2e-3), name
            print("finetune: worst relative gradient-norm deviation %.3g" % worst)
    np.testing.assert_allclose(losses, g["finetune/losses"], rtol=2e-3, atol=1e-5)
    sd = ens.state_dict()
    for k in g.files:
        if k.startswith("finetune/final/") and not k.endswith(NOISY):
            atol = 1e-2 if k.endswith("running_mean") else 4e-3
            # 4 Adam steps at lr 1e-3 on every CNN weight move the embeddings by O(1); statistics of the
            # (large, ~2e3) embeddings are compared to 5e-3 relative
            np.testing.assert_allclose(sd[k[len("finetune/final/"):]].cpu().numpy(), g[k], rtol=5e-3, atol=atol, err_msg=k)
    ens.eval()
    ev = ens(mk.synth_bags(999, 4)[0].cuda())
    np.testing.assert_allclose(ev.cpu().numpy(), g["finetune/eval_after"], rtol=0, atol=2e-2)


def test_train_model_checkpoint_resume_and_report(tmp_path, mk, W):
    """The shell around the step (train.py:53-179, :182-262): epoch loop, checkpoint at each validation improvement,
    resume (weights + Adam moments + epoch + history) == uninterrupted run, bit for bit; test_model's summary agrees
    with sklearn's classification_report."""
    from torch.utils.data import DataLoader, TensorDataset
    T = importlib.import_module(PKG + ".train")

    def loaders():
        x, y = mk.synth_bags(5, 16)
        ds = TensorDataset(torch.as_tensor(x), torch.as_tensor(y))
        return {"train": DataLoader(ds, batch_size=8), "val": DataLoader(ds, batch_size=8), "test": DataLoader(ds, batch_size=8)}

    def run(epochs, path, resume, ens=None):
        ens = ens or build(mk, W)
        for m in ens.modules():                       # dropout off: this test is about bookkeeping, masks are covered above
            if hasattr(m, "p") and m.__class__.__name__ == "Dropout":
                m.p = 0.0
        opt = torch.optim.Adam(T.trainable_params(ens, True), lr=1e-3)
        return T.train_model(ens, loaders(), torch.nn.CrossEntropyLoss(), opt, num_epochs=epochs, patience=None,
                             save_model_path=path, resume=resume)

    full, hist_full, acc_full = run(3, str(tmp_path / "full.pt"), False)
    part, hist_part, _ = run(2, str(tmp_path / "part.pt"), False)
    ck = torch.load(str(tmp_path / "part.pt"), weights_only=True)            # tensors and numbers only
    assert set(ck) == {"epoch", "model", "optimizer", "loss", "accuracy", "history"}
    if ck["epoch"] == 1:                                                     # the last epoch improved: resume continues from it
        resumed, hist_res, acc_res = run(3, str(tmp_path / "part.pt"), True)
        assert hist_res[:2] == hist_part
        if hist_full[2] > max(hist_full[:2]):                                # epoch 3 is the best of both runs -> same final weights
            for (k, a), (_, b) in zip(full.state_dict().items(), resumed.state_dict().items()):
                assert torch.equal(a, b), k
            assert acc_res == acc_full
    assert os.path.exists(str(tmp_path / "full_final.pt"))
    loaded = T.load_model(dict(input_conf="repeat", cnn_conf=dict(mk.CNN_CONF), model_conf=[2, 1], device=torch.device("cuda")),
                          str(tmp_path / "full_final.pt")).cuda()
    acc, results = T.test_model(loaded, loaders()["test"])
    assert acc == acc_full
    # summary vs sklearn on the same predictions
    from sklearn.metrics import classification_report
    _, _, preds, trues = T._evaluate(loaded, loaders()["test"], torch.device("cuda"), collect=True)
    ref = classification_report(trues, preds, labels=list(range(10)), target_names=T.TARGET_NAMES, output_dict=True, zero_division=0)
    for name in T.TARGET_NAMES + ["macro avg", "weighted avg"]:
        for key in ("precision", "recall", "f1-score"):
            want = ref[name][key]
            assert abs(results[name][key] - want) < 1e-9, (name, key)
    assert abs(results["accuracy"] - acc) < 1e-12
