"""Drop-in for the hot-path part of the reference's ``train.py``: the inner training step
(train.py:119-142), ``trainable_params`` (train.py:283-303), the Adam / CrossEntropyLoss choice
(train.py:369-372), ``H5Loader`` (train.py:31-48) and the shell around them: the
``train_model`` epoch loop with early stopping, best-weights restore, checkpoint / resume, ``test_model`` with the
classification summary, ``save_model`` / ``load_model``. Checkpoints hold tensors and numbers only (the reference
pickles whole objects, train.py:249-262); the confusion-matrix plot is not reproduced.

``TrainStep`` fuses zero_grad -> forward -> CrossEntropyLoss -> backward -> Adam into HIP kernel
calls; under ``torch.distributed`` it shards nothing itself (each rank feeds its own bags) and
exchanges (a) BatchNorm statistics (SyncBN, so the result equals the reference's single-process
step on the global batch) and (b) ONE flat gradient buffer with an RCCL all-reduce over xGMI.
"""

import os
import time

import numpy as np
import torch
import torch.nn as nn
from torch.utils.data import Dataset

from . import cnn_train, mla_train, ops
from .params import *  # noqa: F401,F403


class H5Loader(Dataset):
    """Dataset over array-likes (HDF5 datasets or ndarrays): train.py:31-48."""

    def __init__(self, X_desc, y_desc):
        self.X_desc, self.y_desc = X_desc, y_desc

    def __len__(self):
        return self.y_desc.shape[0]

    def __getitem__(self, idx):
        return (self.X_desc[idx], self.y_desc[idx])


def trainable_params(model, feature_extract):
    """train.py:283-303: the parameters to hand to the optimizer (prints their names)."""
    params_to_update = model.parameters()
    print("Params to learn:")
    if feature_extract:
        params_to_update = []
        for name, param in model.named_parameters():
            if param.requires_grad:
                params_to_update.append(param)
                print("\t", name)
    else:
        for name, param in model.named_parameters():
            if param.requires_grad:
                print("\t", name)
    return params_to_update


class TrainStep:
    """One fused training step of an ``Ensemble``: frozen CNN (the reference default
    ``cnn_trainable=False``; model.py:159-160) or finetune (every parameter trainable, train.py:96-97;
    f32 precision).

    All trainable parameters are re-seated as views of ONE flat float32 buffer (same for
    gradients and the two Adam moments), so the data-parallel exchange is a single all-reduce
    and Adam a single kernel. ``attention_modules.*.fcf.*`` never receive a gradient
    (model.py:237-238); like torch's Adam (which skips ``grad is None``) they are left untouched.
    """

    def __init__(self, clf, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, process_group=None, params=None, sync_bn=True, graph=None):
        """``params``: the parameters the caller's optimizer holds (``optimizer.param_groups[...]["params"]``); None =
        every parameter that requires grad. Exactly the tensors in ``params`` that require grad are updated, as
        ``optimizer.step()`` does in the reference (train.py:138; torch's Adam skips parameters whose ``.grad`` is None,
        i.e. the frozen ones). Gradients of parameters OUTSIDE that set are not computed: the reference computes and then
        never applies them (its ``__main__`` builds the Adam before ``set_requires_grad(clf, True)``, train.py:369-370 vs
        :96-97, so its "finetune" run only ever steps the MLA head) -- the updated model is the same."""
        self.clf, self.lr, self.betas, self.eps, self.t = clf, lr, betas, eps, 0
        # sync_bn=False: per-shard BatchNorm statistics under data parallelism (DistributedDataParallel semantics) instead of the
        # global-batch statistics that reproduce the reference's single-process step; see ops.Dist
        self.dist = ops.Dist(process_group, sync_bn=sync_bn)
        self.exposed = None                                 # bench.py: list of (event, event) around the wait for the comm stream
        # use_graph: replay the whole step as one HIP graph from the second step of a batch shape on (single process only; env
        # MLA_TRAIN_GRAPH=0 or graph=False keeps every step eager)
        self.use_graph = (os.environ.get("MLA_TRAIN_GRAPH", "1") == "1") if graph is None else bool(graph)
        self._graph, self._eager_shape, self._dev_t = None, None, -1
        held = None if params is None else {id(p) for p in params}
        named = [(n, p) for n, p in clf.named_parameters()
                 if p.requires_grad and ".fcf." not in n and (held is None or id(p) in held)]
        if not named:
            raise ValueError("no trainable parameter to update")
        # CNN gradients are needed iff the update set holds a CNN parameter; any subset of the CNN is fine (the backward
        # pass stops at the lowest layer that needs a gradient and skips the weight gradients nobody asked for)
        self.finetune = any(n.startswith("cnn.") for n, _ in named)
        if self.finetune and clf.cnn.precision not in ("f32", "bf16"):
            raise NotImplementedError("CNN gradients are built for precision 'f32' (exact) and 'bf16' (bf16 arithmetic, f32 "
                                      "master weights), not %r" % clf.cnn.precision)
        dev = named[0][1].device
        assert dev.type == "cuda", "TrainStep needs the model on the GPU (train_model moves it there, train.py:101)"
        pad4 = lambda k: (k + 3) // 4 * 4                 # every tensor starts 16-byte aligned (GEMM operand rule)
        total = sum(pad4(p.numel()) for _, p in named)
        self.n_params = sum(p.numel() for _, p in named)
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(total, dtype=torch.float32, device=dev)
        self.step_dev = torch.zeros(1, dtype=torch.int64, device=dev)       # completed steps: the dropout call counter of the graph-captured step
        self.adam_scal = torch.zeros(2, dtype=torch.float32, device=dev)    # Adam's step-dependent scalars, written in front of each replay
        self.grads, self._seated, off = {}, [], 0
        spans = {}                                        # bucket id -> [first float, one past the last] of the flat buffers
        for n, p in named:
            k = p.numel()
            self.flat_p[off:off + k].copy_(p.detach().reshape(-1))
            p.data = self.flat_p[off:off + k].view(p.shape)
            self.grads[n] = self.flat_g[off:off + k].view(p.shape)
            self._seated.append((n, p, self.flat_p.data_ptr() + 4 * off))
            b = self._bucket_of(n)
            spans[b] = [min(spans.get(b, [off, 0])[0], off), off + pad4(k)]
            off += pad4(k)
        # Gradient buckets of the data-parallel exchange: contiguous ranges of the flat gradient buffer that become final
        # at known points of the backward pass -- the head first, then the CNN from its last Linear down to conv1 -- so that
        # each range can be all-reduced on a second HIP stream while the layers below are still computing (the parameters
        # are laid out in named_parameters() order = forward order, so "the layers from position p upwards" is one range):
        #   "mla" 3.2 MB | "fc12" embeddings.2 + .4: 69 MB | "fc0" embeddings.0: 201 MB | "conv56" 14 MB | "conv14" 4 MB
        # xGMI rings are per-link bound, so a few large messages beat many small ones; the order above is the order in which
        # the backward pass finishes them.
        self.buckets = {b: tuple(v) for b, v in spans.items()}
        self.overlap = os.environ.get("MLA_DIST_OVERLAP", "1") == "1"
        self._comm_stream = None
        # MLA parameters outside the update set still need somewhere to write their gradient (the head's backward always
        # runs whole: it is 0.3 ms); those scratch tensors are never read
        self.mla_grads = {}
        for n, p in clf.mla.named_parameters():
            if ".fcf." in n:
                continue
            self.mla_grads[n] = self.grads["mla." + n] if "mla." + n in self.grads else torch.empty_like(p, dtype=torch.float32, device=dev)

    # layer position (cnn_train.backward's `pos`: 0..5 conv1..conv6, 6..8 the three Linear layers) at which a bucket is final
    BUCKET_TRIGGER = {"fc12": 7, "fc0": 6, "conv56": 4, "conv14": 0}

    @staticmethod
    def _bucket_of(name):
        if not name.startswith("cnn."):
            return "mla"
        key = name.split(".")[-3:-1]                      # (..., "features" | "embeddings" | "0", index, "weight" | "bias")
        idx = int(key[1])
        if key[0] == "embeddings":
            return "fc0" if idx == 0 else "fc12"
        return "conv56" if idx >= 11 else "conv14"

    def _reduce_bucket(self, name):
        """All-reduce one finished gradient range on the communication stream (it first waits for the compute stream's work
        enqueued so far, i.e. for the kernels that produced the range)."""
        a, b = self.buckets[name]
        cur = torch.cuda.current_stream()
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=self.flat_g.device)
        self._comm_stream.wait_stream(cur)
        with torch.cuda.stream(self._comm_stream):
            self.dist.all_reduce_sum(self.flat_g[a:b], "grad:" + name)

    def state_dict(self):
        """Optimizer state as plain tensors (Adam moments over the flat buffer + step count): what
        ``optimizer.state_dict()`` holds in the reference's checkpoints (train.py:249-258)."""
        return {"adam_m": self.flat_m.clone(), "adam_v": self.flat_v.clone(), "step": torch.tensor(self.t),
                "lr": torch.tensor(self.lr), "betas": torch.tensor(self.betas), "eps": torch.tensor(self.eps)}

    def load_state_dict(self, sd):
        assert sd["adam_m"].numel() == self.flat_m.numel(), "checkpoint was written for a different set of trainable parameters"
        self.flat_m.copy_(sd["adam_m"]); self.flat_v.copy_(sd["adam_v"]); self.t = int(sd["step"])
        self.lr, self.eps = float(sd["lr"]), float(sd["eps"])
        self.betas = tuple(float(b) for b in sd["betas"])

    def __call__(self, inputs, labels):
        """inputs (B, T, 1, 96, 64) (or whatever ``clf.input`` reshapes), labels (B,) int64.
        Returns (loss, hits) as device tensors (no host sync; train.py:141-142 syncs via .item()): hits = int32
        [n_correct, n_labels_out_of_range] over the global batch (``ops.raise_on_bad_labels(hits)`` -> n_correct or IndexError).
        With the HIP graph in use the two tensors (and ``last_out``) are the graph's own buffers: valid until the next step."""
        clf = self.clf
        for n, p, ptr in self._seated:
            if p.data_ptr() != ptr:
                raise RuntimeError("parameter %s no longer lives in the flat buffer of this TrainStep (the model was moved or "
                                   "cast after the step was built): build a new TrainStep" % n)
        clf.train()
        ops.check_labels(labels, clf.num_classes)
        if self._graph_usable(inputs):
            return self._graphed(inputs, labels)
        with torch.no_grad():
            loss, hits, out = self._body(inputs, labels.to(inputs.device).long().contiguous(), captured=False)
        self._eager_shape = tuple(inputs.shape)
        self.last_out = out
        return loss, hits

    # ---- the step as ONE HIP graph (no data-parallel group): ~130 small launches of the head's forward / backward / Adam and the
    # CNN's kernels become one graph launch; what the host used to pass per step -- Adam's step count, the dropout call number --
    # is read from device memory instead (mla_adam_prepare + mla_adam_step_dev, mla_dropout_mask_dev), so a replay is a real next step.

    def _dropouts(self):
        return [m for m in self.clf.mla.modules() if type(m).__name__ == "Dropout"]

    def _graph_usable(self, inputs):
        if not self.use_graph or self.dist.active or ops.profile is not None or not inputs.is_cuda or inputs.dtype != torch.float32:
            return False
        if not inputs.is_contiguous() or any(d.mask is not None for d in self._dropouts()):
            return False                      # injected masks change per step on the host: the eager path
        # first step of a shape runs eagerly: it sizes the library's workspaces and warms the weight caches outside any capture
        return self._eager_shape == tuple(inputs.shape)

    def _graphed(self, inputs, labels):
        dev = inputs.device
        drops = self._dropouts()
        g = self._graph
        if g is not None and (g["shape"] != tuple(inputs.shape) or any(d.calls - self.t != g["bases"][id(d)] for d in drops)):
            g = self._graph = None                                 # another batch size, or somebody else drew masks in between
        if self._dev_t != self.t:
            self.step_dev.fill_(self.t)
            self._dev_t = self.t
        if g is None:
            g = {"shape": tuple(inputs.shape), "x": inputs, "y": labels.to(dev).long().contiguous().clone(),
                 "bases": {id(d): d.calls - self.t for d in drops}}
            torch.cuda.synchronize()
            g["graph"] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g["graph"], capture_error_mode="relaxed"), torch.no_grad():
                g["loss"], g["hits"], g["out"] = self._body(g["x"], g["y"], captured=True, bases=g["bases"])
            self._graph = g
        else:
            if inputs.data_ptr() != g["x"].data_ptr():
                g["x"].copy_(inputs, non_blocking=True)
            g["y"].copy_(labels.to(dev, non_blocking=True).long(), non_blocking=True)
        ops.adam_prepare(self.adam_scal, self.lr, self.betas[0], self.betas[1], self.t + 1)
        g["graph"].replay()
        self.t += 1
        self._dev_t = self.t
        for d in drops:
            d.calls += 1
        if self.finetune:                          # the graph re-derives its own weight copies; anything cached outside it is stale
            for m in self.clf.cnn.modules():
                if hasattr(m, "_cache"):
                    m._cache.key = None
        self.last_out = g["out"]
        return g["loss"], g["hits"]

    def _body(self, inputs, labels, captured, bases=None):
        """The kernel sequence of one step (train.py:124-138). captured: being recorded into the HIP graph."""
        clf = self.clf
        B_global = inputs.shape[0] * self.dist.world
        x = clf.input(inputs)
        if self.finetune:
            feats, cnn_tape = cnn_train.forward(clf.cnn.cnn_model, x, clf.cnn.precision)
        else:
            feats = clf.cnn(x)
        ctx = mla_train.Ctx(tape=True, dist=self.dist, counter=self.step_dev if captured else None, bases=bases)
        out = mla_train.mla_forward(clf.mla, feats.reshape(-1, T, clf.emb_input_size), ctx)
        loss, dout, hits = ops.cross_entropy(out, labels, 1.0 / B_global)
        d_feats = mla_train.mla_backward(clf.mla, ctx, dout, self.mla_grads, need_input_grad=self.finetune)
        bucketed = self.dist.active and self.finetune and self.overlap
        if bucketed:
            # the head's gradients are final: reduce them while the CNN backward runs; each CNN bucket follows as soon as
            # the lowest layer it holds is done. No other collective is issued until the compute stream has waited for
            # the communication stream below, so the two streams never use the communicator at the same time.
            pending = [b for b in ("mla", "fc12", "fc0", "conv56", "conv14") if b in self.buckets]
            if "mla" in pending:
                self._reduce_bucket("mla"); pending.remove("mla")

            def after_layer(pos):
                for b in list(pending):
                    if self.BUCKET_TRIGGER[b] == pos:
                        self._reduce_bucket(b); pending.remove(b)
            cnn_train.backward(clf.cnn.cnn_model, cnn_tape, d_feats, self.grads, "cnn.cnn_model.", after_layer)
            for b in pending:                          # buckets whose trigger layer lies below the lowest trained layer
                self._reduce_bucket(b)
            if self.exposed is not None:               # how long the compute stream really waits for the exchange
                e0, e1 = ops._event(), ops._event()
                e0.record()
                torch.cuda.current_stream().wait_stream(self._comm_stream)
                e1.record()
                self.exposed.append((e0, e1))
            else:
                torch.cuda.current_stream().wait_stream(self._comm_stream)
        elif self.finetune:
            cnn_train.backward(clf.cnn.cnn_model, cnn_tape, d_feats, self.grads, "cnn.cnn_model.")
        if self.dist.active:
            if not bucketed:
                self.dist.all_reduce_sum(self.flat_g, "grad:flat")
            self.dist.all_reduce_sum(loss, "loss")
            self.dist.all_reduce_sum(hits, "hits")             # [running_corrects (train.py:142), #bad labels] over the global batch
        if captured:
            ops.adam_step_dev(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.betas[0], self.betas[1], self.eps, self.adam_scal, self.step_dev)
        else:
            self.t += 1
            ops.adam_step(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.lr, self.betas[0], self.betas[1], self.eps, self.t)
            if self.finetune:                      # derived (repacked / bf16) weight copies are stale now
                for m in clf.cnn.modules():
                    if hasattr(m, "_cache"):
                        m._cache.key = None
        return loss, hits, out


def _evaluate(clf, loader, device, collect=False):
    clf.eval()
    tot_loss, tot_hits, n, preds, trues = 0.0, 0, 0, [], []
    with torch.no_grad():
        for inputs, labels in loader:
            host_labels = labels
            inputs, labels = inputs.to(device).float(), labels.to(device).long()
            out = clf(inputs)
            ops.check_labels(host_labels, out.shape[1])
            loss, _, hits = ops.cross_entropy(out, labels.contiguous(), 1.0 / inputs.shape[0], want_grad=False)
            tot_loss += float(loss) * inputs.shape[0]
            tot_hits += ops.raise_on_bad_labels(hits)
            n += inputs.shape[0]
            if collect:
                preds.append(torch.max(out, 1)[1].cpu()); trues.append(labels.cpu())
    if collect:
        return tot_loss / max(n, 1), tot_hits / max(n, 1), torch.cat(preds), torch.cat(trues)
    return tot_loss / max(n, 1), tot_hits / max(n, 1)


def classification_summary(y_true, y_pred, target_names=TARGET_NAMES):
    """Per-class precision / recall / f1 / support + accuracy and macro / weighted averages, laid out like
    sklearn's ``classification_report(output_dict=True)`` (train.py:246), and the row-normalised confusion
    matrix in percent (train.py:237-241). Plain numpy: nothing here is on the GPU path."""
    y_true, y_pred = np.asarray(y_true), np.asarray(y_pred)
    k = len(target_names)
    cm = np.zeros((k, k), dtype=np.float64)
    np.add.at(cm, (y_true, y_pred), 1.0)
    tp, support, predicted = np.diag(cm), cm.sum(axis=1), cm.sum(axis=0)
    with np.errstate(divide="ignore", invalid="ignore"):
        prec = np.where(predicted > 0, tp / predicted, 0.0)
        rec = np.where(support > 0, tp / support, 0.0)
        f1 = np.where(prec + rec > 0, 2 * prec * rec / (prec + rec), 0.0)
        cm_pct = np.where(support[:, None] > 0, cm * 100.0 / support[:, None], 0.0).astype(np.float32)
    res = {name: {"precision": float(prec[i]), "recall": float(rec[i]), "f1-score": float(f1[i]), "support": int(support[i])}
           for i, name in enumerate(target_names)}
    total = float(support.sum())
    res["accuracy"] = float(tp.sum() / total) if total else 0.0
    for tag, wts in (("macro avg", np.ones(k)), ("weighted avg", support)):
        w = wts / wts.sum() if wts.sum() else wts
        res[tag] = {"precision": float((prec * w).sum()), "recall": float((rec * w).sum()), "f1-score": float((f1 * w).sum()),
                    "support": int(total)}
    return res, cm_pct


def test_model(model, dataloader, criterion=None, optimizer=None):
    """train.py:182-247: final test pass -> (test accuracy, classification summary dict). The confusion-matrix
    plot of the reference is not reproduced; the matrix itself is returned under ``results["confusion_matrix_pct"]``."""
    if dataloader is None:
        return None
    device = next(model.parameters()).device
    since = time.time()
    print("Testing"); print("-" * 10)
    loss, acc, preds, trues = _evaluate(model, dataloader, device, collect=True)
    print("{} Loss: {:.4f}, Acc: {:.4f}".format("test", loss, acc))
    print("Testing complete in {:.0f}s".format(time.time() - since))
    results, cm = classification_summary(trues.numpy(), preds.numpy())
    results["confusion_matrix_pct"] = cm
    return acc, results


def _save_checkpoint(model, step, epoch, loss, accuracy, history, path):
    """train.py:249-258 with tensors and numbers only (the reference pickles whole objects; these files load with
    ``torch.load(weights_only=True)``): model ``state_dict`` + Adam state of the fused step + bookkeeping."""
    torch.save({"epoch": int(epoch), "model": model.state_dict(), "optimizer": step.state_dict(), "loss": float(loss),
                "accuracy": float(accuracy), "history": [float(h) for h in history]}, path)


def _resume_from_checkpoint(path, model, step):
    """train.py:260-262: restores `model` and `step` in place, returns (epoch, loss, accuracy, history)."""
    d = torch.load(path, map_location=next(model.parameters()).device, weights_only=True)
    model.load_state_dict(d["model"])
    step.load_state_dict(d["optimizer"])
    return d["epoch"], d["loss"], d["accuracy"], list(d["history"])


def save_model(model, path):
    """train.py:265-266."""
    torch.save(model.state_dict(), path)


def load_model(model_args, path):
    """train.py:268-271: build an Ensemble from its constructor arguments and load a ``save_model`` file."""
    from .model import Ensemble
    m = Ensemble(**model_args)
    m.load_state_dict(torch.load(path, map_location="cpu", weights_only=True))
    return m


def train_model(clf, dataloaders, criterion, optimizer, num_epochs=25, patience=10, save_model_path=None, resume=False,
                finetune=False):
    """Epoch loop with the reference's signature and return value (train.py:53-179):
    (model with the best-validation weights, validation accuracy history, test_model's result).
    ``criterion`` must be nn.CrossEntropyLoss and ``optimizer`` a one-group torch.optim.Adam without weight decay /
    amsgrad: its hyper-parameters AND its parameter list are read -- the fused HIP step updates exactly the tensors the
    optimizer holds (and that require grad), as ``optimizer.step()`` does; the torch optimizer object itself is not
    stepped (its ``state`` stays empty; the moments live in the checkpoint written here). Checkpoints hold tensors only
    (``_save_checkpoint``); ``resume=True`` continues from ``save_model_path`` (epoch, best accuracy, history,
    weights and Adam moments), as train.py:80-94 does from its pickled objects."""
    if not isinstance(criterion, nn.CrossEntropyLoss) or not isinstance(optimizer, torch.optim.Adam):
        raise TypeError("the HIP training step implements CrossEntropyLoss + Adam (train.py:369-372)")
    if len(optimizer.param_groups) != 1:
        raise NotImplementedError("the HIP Adam step takes ONE parameter group (train.py:369-370 builds one)")
    g = optimizer.param_groups[0]
    if g.get("weight_decay", 0) != 0 or g.get("amsgrad", False) or g.get("maximize", False):
        raise NotImplementedError("weight_decay / amsgrad / maximize are not implemented by the HIP Adam step (train.py:369 uses none)")
    if finetune:
        from .model import set_requires_grad
        set_requires_grad(clf, True)                       # train.py:96-97 (AFTER the caller built the optimizer)
    import copy
    device = torch.device("cuda", torch.cuda.current_device())
    clf.to(device)                                         # train.py:101
    step = TrainStep(clf, lr=g["lr"], betas=g["betas"], eps=g["eps"], params=g["params"])     # steps what the optimizer holds
    since, val_acc_history, best_acc, best_epoch, first_epoch = time.time(), [], 0.0, 0, 0
    if resume:
        assert save_model_path is not None
        if not os.path.exists(save_model_path):
            raise Exception("No such model file in the specified path.")
        ep, _, best_acc, val_acc_history = _resume_from_checkpoint(save_model_path, clf, step)
        first_epoch, best_epoch = ep + 1, ep
    best_wts = copy.deepcopy(clf.state_dict())
    test_loader = dataloaders.pop("test", None)
    for epoch in range(first_epoch, num_epochs):
        print("Epoch {}/{}".format(epoch + 1, num_epochs)); print("-" * 10)
        run_loss, run_hits, n = 0.0, 0, 0
        for inputs, labels in dataloaders["train"]:
            loss, hits = step(inputs.to(device).float(), labels)
            # under data parallelism the step returns the GLOBAL mean loss and the GLOBAL hit count (train.py:141-142 on the whole batch)
            seen = inputs.shape[0] * (step.dist.world if step.dist.active else 1)
            run_loss += float(loss) * seen; run_hits += ops.raise_on_bad_labels(hits); n += seen
        print("train Loss: {:.4f}, Acc: {:.4f}".format(run_loss / max(n, 1), run_hits / max(n, 1)))
        v_loss, v_acc = _evaluate(clf, dataloaders["val"], device)
        print("val Loss: {:.4f}, Acc: {:.4f}".format(v_loss, v_acc))
        val_acc_history.append(v_acc)
        if v_acc > best_acc:
            best_acc, best_epoch, best_wts = v_acc, epoch, copy.deepcopy(clf.state_dict())
            if save_model_path:
                _save_checkpoint(clf, step, epoch, v_loss, best_acc, val_acc_history, save_model_path)
                print("Model checkpoint saved successfully in the given path!")
        if patience is not None and epoch - best_epoch >= patience:
            break
    print("Training complete in {:.0f}s; best val Acc: {:4f}".format(time.time() - since, best_acc))
    clf.load_state_dict(best_wts)
    tested = test_model(clf, test_loader)
    final = save_model_path or "best_weights.h5"
    root, ext = os.path.splitext(final)
    save_model(clf, root + ("_final_finetuned" if finetune else "_final") + ext)       # train.py:173-177
    return clf, val_acc_history, tested                    # test_model's (accuracy, summary) tuple or None, as train.py:179
