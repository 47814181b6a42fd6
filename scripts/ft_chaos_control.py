"""Control for the bf16 finetune test (tests/test_finetune_bf16_gpu.py): how far do the loss curve and the step-4 scores of the
EXACT f32 path move when its inputs are perturbed by a relative eps (1e-4 ... 3e-3, the size of the bf16 forward deviation)?
Output recorded in profiles/r02_finetune_bf16_control.txt."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
PKG = "audio-classification-using-a-deep-cnn-combined-with-multi-level-attention_amd"
mk = importlib.import_module("make_golden"); W = importlib.import_module(PKG + ".weights")
M = importlib.import_module(PKG + ".model"); TR = importlib.import_module(PKG + ".train")
g = np.load(os.path.join(ROOT, "tests/golden/train.npz"))
for eps in (0.0, 1e-4, 1e-3, 3e-3):
    ens = M.Ensemble("repeat", dict(mk.CNN_CONF), [2, 1], torch.device("cuda"), precision="f32")
    ens.load_state_dict({k: torch.as_tensor(v) for k, v in W.make_state_dict(7, W.ensemble_shapes((2, 1), False)).items()})
    ens.cuda(); M.set_requires_grad(ens, True)
    step = TR.TrainStep(ens, lr=1e-3)
    losses = []
    for s in range(4):
        x, y = mk.synth_bags(100 + s, 4)
        noise = torch.from_numpy(W.uniform(900 + s, 1, x.numel())).reshape(x.shape)
        x = x * (1 + eps * noise)
        masks = mk.make_masks(200 + s, [2, 1], 4)
        for lvl, em in enumerate(ens.mla.embedded_mappings):
            for j, d in enumerate(em.dropouts):
                d.mask = masks["mla.embedded_mappings.%d.dropouts.%d" % (lvl, j)]
        loss, _ = step(x.cuda(), y.cuda())
        losses.append(float(loss))
    d = np.abs(step.last_out.cpu().numpy() - g["finetune/out_last"])
    print("input perturbation %.0e: loss dev %.3g, step-4 score deviation mean %.3g max %.3g"
          % (eps, np.abs(np.array(losses) / g["finetune/losses"] - 1).max(), d.mean(), d.max()))
