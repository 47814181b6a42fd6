"""Drop-in ``torchvggish`` package (mel_features, vggish_input, vggish_params, vggish)."""
