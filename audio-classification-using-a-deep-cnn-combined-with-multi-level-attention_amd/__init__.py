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
