"""Importable alias for the hyphen-named package directory (``import mla_amd``)."""
import importlib
import sys

sys.modules[__name__] = importlib.import_module(
    "audio-classification-using-a-deep-cnn-combined-with-multi-level-attention_amd")
