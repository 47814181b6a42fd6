"""Bit-level comparison of two library builds on the same seeded input: each build runs the bf16 wave -> logits forward of 64 bags in
its own process (the library is bound once per process) and dumps logits + embeddings; the parent compares the dumps.
    python scripts/compare_variants.py build/variants/libmla_notall.so main"""
import importlib, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "audio-classification-using-a-deep-cnn-combined-with-multi-level-attention_amd"

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch
    W = importlib.import_module(PKG + ".weights")
    M = importlib.import_module(PKG + ".model")
    conf = dict(cnn_type="vggish", num_classes=10, use_pretrained=False, just_bottlenecks=False, cnn_trainable=False,
                first_cnn_layer_trainable=False, in_channels=1)
    out = {}
    for prec in ("bf16", "bf16x3"):
        ens = M.Ensemble("repeat", conf, [2, 1], torch.device("cuda"), precision=prec)
        ens.load_state_dict({k: torch.as_tensor(v) for k, v in W.make_state_dict(7, W.ensemble_shapes((2, 1), False)).items()})
        ens.cuda().eval()
        pcm = torch.from_numpy(W.uniform(3, 1, 64 * 160000, lo=-1.0, hi=1.0)).reshape(64, 160000).cuda()
        with torch.no_grad():
            out[prec] = ens.forward_waveforms(pcm).float().cpu().numpy()
    np.savez(sys.argv[2], **out)
    sys.exit(0)

dumps = []
for i, lib in enumerate(sys.argv[1:3]):
    env = dict(os.environ)
    if lib != "main":
        env["MLA_HIP_LIB"] = os.path.join(ROOT, lib)
    path = "/tmp/_cmp_%d.npz" % i
    subprocess.run([sys.executable, __file__, "--child", path], env=env, check=True)
    dumps.append(np.load(path))
for k in dumps[0].files:
    a, b = dumps[0][k], dumps[1][k]
    print(k, "bit-identical" if np.array_equal(a.view(np.uint32), b.view(np.uint32)) else "DIFFERENT max|d| %.3e" % np.abs(a - b).max(), a.shape)
