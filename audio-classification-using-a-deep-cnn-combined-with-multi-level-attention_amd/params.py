"""Model constants consumed by the hot path.

Mirrors the model section of the reference's ``params.py:24-32`` (``T, M_VGGISH,
M_VGGISH_JB, H, DR, K, S_VGGISH_SHAPE``) and the class labels (``params.py:15-16``: UrbanSound8K's ten
classes in classID order, used by ``train.test_model``'s report). The dataset paths and the
ResNet constants of the reference are outside SURVEY.md section 8 and are not
reproduced. ``model.py`` binds these at import time exactly as the reference's
``from params import *`` does (``model.py:9``), so they are compile-time
constants of the model and of the HIP kernels instantiated for it.
"""

S_VGGISH_SHAPE = (96, 64)     # one VGGish example: 96 STFT frames x 64 mel bands
T = 10                        # examples (0.96 s clips) per bag
M_VGGISH = 128                # VGGish embedding width
M_VGGISH_JB = 512 * 6 * 4     # flattened conv bottleneck width (just_bottlenecks=True)
H = 600                       # hidden width of the embedded mappings
DR = 0.4                      # dropout rate inside the embedded mappings
K = 10                        # classes
TARGET_NAMES = ["air_conditioner", "car_horn", "children_playing", "dog_bark", "drilling", "engine_idling", "gun_shot",
                "jackhammer", "siren", "street_music"]

BATCH_SIZE = 8
NUM_EPOCHS = 25
FEATURE_EXTRACT = True
LR = 0.001
