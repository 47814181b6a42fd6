"""VGGish front-end / embedding constants, re-exported under the reference's names.

Same names and values as the reference's ``torchvggish/vggish_params.py:22-53``;
the HIP log-mel kernel (csrc/logmel.hip) is compiled for exactly this
configuration (16 kHz, 400/160/512 STFT, 64 HTK mel bands 125-7500 Hz, log
offset 0.01, 96-frame examples with hop 96).
"""

NUM_FRAMES, NUM_BANDS, EMBEDDING_SIZE = 96, 64, 128

SAMPLE_RATE = 16000
STFT_WINDOW_LENGTH_SECONDS, STFT_HOP_LENGTH_SECONDS = 0.025, 0.010
NUM_MEL_BINS = NUM_BANDS
MEL_MIN_HZ, MEL_MAX_HZ = 125, 7500
LOG_OFFSET = 0.01
EXAMPLE_WINDOW_SECONDS = EXAMPLE_HOP_SECONDS = 0.96

PCA_EIGEN_VECTORS_NAME, PCA_MEANS_NAME = "pca_eigen_vectors", "pca_means"
QUANTIZE_MIN_VAL, QUANTIZE_MAX_VAL = -2.0, +2.0

INIT_STDDEV = 0.01
LEARNING_RATE = 1e-4
ADAM_EPSILON = 1e-8

INPUT_OP_NAME = "vggish/input_features"
INPUT_TENSOR_NAME = INPUT_OP_NAME + ":0"
OUTPUT_OP_NAME = "vggish/embedding"
OUTPUT_TENSOR_NAME = OUTPUT_OP_NAME + ":0"
AUDIO_EMBEDDING_FEATURE_NAME = "audio_embedding"
