"""Drop-in for the hot-path-adjacent part of the reference's ``dataset.py``:
the native spectrogram path of ``create_spec`` (dataset.py:318-324) and ``split`` /
``overlapping_split`` / ``contiguous_split`` (dataset.py:329-363), i.e. the actual caller of
``waveform_to_examples`` in the reference's data pipeline. Same names and argument order; tensors
stay on the GPU. The librosa path, the UrbanSound8K fold logic and the HDF5 writer are outside scope.

``clips_to_frames`` is the batched fast path: PCM of many clips -> (clips, T, 1, 64, 96) in two
kernels (fused log-mel + the re-framing gather), which is the tensor ``Input`` reshapes at
model.py:98-99 (a reshape, not a transpose -- reproduced as is)."""

import ctypes

import numpy as np
import torch

from . import _lib, frontend
from .params import T


def _frames(examples, clips, ex_per_clip, n_frames, frame_len, stride):
    out = torch.empty((clips, n_frames, 64, frame_len), dtype=torch.float32, device=examples.device)
    vp = ctypes.c_void_p
    _lib.check(_lib.lib().mla_dataset_frames(vp(examples.data_ptr()), clips, ex_per_clip, n_frames, frame_len, stride,
                                             vp(out.data_ptr()), _lib.stream_ptr()))
    return out


def create_spec(audio_array, cnn_type, sr, samples_num, x_size, y_size, use_librosa, overlap):
    """(64, 384) spectrogram of one clip on the GPU (native VGGish path only)."""
    if use_librosa or cnn_type != "vggish":
        raise NotImplementedError("only the native VGGish path (use_librosa=False) is on the HIP hot path")
    ex = frontend.waveforms_to_examples(frontend.as_device_mono(audio_array)[None])
    if ex.shape[0] > 4:
        raise ValueError("could not broadcast input array from shape (%d,96,64) into shape (4,96,64)" % ex.shape[0])
    return _frames(ex, 1, ex.shape[0], 1, 384, 0)[0, 0]


def overlapping_split(spec, num_frames, frame_length):
    """Zero-copy strided view (num_frames, 64, frame_length), stride (W - frame_length) // (num_frames - 1)."""
    stride = (spec.shape[1] - frame_length) // (num_frames - 1)
    return spec.unfold(1, frame_length, stride).permute(1, 0, 2)[:num_frames]


def contiguous_split(spec, num_frames, frame_length):
    return spec.unfold(1, frame_length, frame_length).permute(1, 0, 2)[:num_frames]


def split(spec, num_frames, x_size, y_size, overlap):
    frames = overlapping_split(spec, num_frames, x_size) if overlap else contiguous_split(spec, num_frames, x_size)
    assert tuple(frames.shape[1:]) == (y_size, x_size)
    return frames


def clips_to_frames(pcm, overlap=True):
    """(clips, n_samples) device PCM (<= 4 s per clip) -> (clips, T, 1, 64, 96) float32, the layout
    load_hdf5 stores (dataset.py:252-255) and Ensemble consumes."""
    clips = pcm.shape[0]
    ex = frontend.waveforms_to_examples(pcm)
    per = ex.shape[0] // max(clips, 1)
    if per > 4:
        raise ValueError("clips longer than 4 examples do not fit the 4-slot spectrogram (dataset.py:321-322)")
    n = T if overlap else 4
    stride = (384 - 96) // (T - 1) if overlap else 96
    return _frames(ex, clips, per, n, 96, stride)[:, :, None]
