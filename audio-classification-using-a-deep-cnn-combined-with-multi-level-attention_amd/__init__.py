"""MI355X-native hot path of caesar-one/audio-classification-using-a-deep-cnn-combined-
with-multi-level-attention: waveform -> log-mel examples -> VGGish -> multi-level
attention -> class scores, as hand-written gfx950 HIP kernels behind a C ABI
(``include/mla_hip.h``), with the reference's Python API surface on top:

    <pkg>.torchvggish.{mel_features, vggish_input, vggish_params, vggish}
    <pkg>.model    (Ensemble, Input, CNN, CnnFlatten, EmbeddedMapping, AttentionModule,
                    MultiLevelAttention, set_requires_grad)
    <pkg>.params, <pkg>.train (the training step), <pkg>.dataset (native spectrogram re-framing)

The directory name is not a Python identifier; import it with
``importlib.import_module("audio-classification-using-a-deep-cnn-combined-with-multi-level-attention_amd")``
or through the ``mla_amd`` alias module at the repo root. ``install_dropin()``
additionally registers ``torchvggish``, ``model``, ``params``, ``train`` and ``dataset`` as top-level module
names, which is how the reference's own scripts import them (model.py:7-9).

Importing this package does not load the HIP library; the first kernel call does
(``_lib.lib()``), and fails loudly if ``libmla_hip.so`` has not been built.
"""

import importlib
import sys

__all__ = ["install_dropin", "PKG_NAME"]
PKG_NAME = __name__


def install_dropin():
    """Expose the drop-in modules under the reference's top-level import names."""
    for short in ("params", "torchvggish", "model", "train", "dataset"):
        sys.modules[short] = importlib.import_module(__name__ + "." + short)
    for sub in ("mel_features", "vggish_input", "vggish_params", "vggish"):
        sys.modules["torchvggish." + sub] = importlib.import_module(__name__ + ".torchvggish." + sub)


def _smoke_model():
    """Second half of __graft_entry__.smoke(): one bag, wave -> scores, HIP vs oracle."""
    import numpy as np
    import torch
    from oracle import frontend as ofe
    from oracle import model as omodel
    W = importlib.import_module(__name__ + ".weights")
    M = importlib.import_module(__name__ + ".model")
    conf = dict(cnn_type="vggish", num_classes=10, use_pretrained=False, just_bottlenecks=False,
                cnn_trainable=False, first_cnn_layer_trainable=False, in_channels=1)
    sd = W.make_state_dict(6, W.ensemble_shapes((2, 1), False))
    ens = M.Ensemble("repeat", conf, [2, 1], torch.device("cuda"))
    ens.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    ens.cuda().eval()
    wav = W.waveform(5, 160000, 2)
    ref = omodel.ensemble_forward(omodel.to_torch(sd), torch.as_tensor(ofe.batch_examples(wav.astype(np.float64))).float())
    for prec, tol in (("f32", 1e-4), ("bf16", 5e-2)):
        got = ens.set_precision(prec).forward_waveforms(torch.from_numpy(wav).cuda()).cpu().numpy()
        err = float(np.abs(got - ref.numpy()).max() / np.abs(ref.numpy()).max())
        print("smoke: wave->scores %s rel err vs oracle = %.3g" % (prec, err))
        assert err < tol, (prec, err)
