"""ORACLE (test infrastructure, not product code): numpy restatement of the reference's native
spectrogram re-framing, dataset.py:318-324 (create_spec, use_librosa=False) and :329-363 (split,
overlapping_split, contiguous_split). Pinned by tests/golden/dataset.npz, produced by the reference's
own functions (imported with stubs for librosa / soundfile / h5py, which that path never calls)."""

import numpy as np

from . import frontend


def create_spec_native(audio_array, sr=16000):
    """dataset.py:318-324 -- <= 4 examples zero-padded (with 0.0, not log 0.01) to 4 slots, each
    transposed to (64, 96), concatenated along time -> (64, 384)."""
    slots = frontend.waveform_to_examples(audio_array, sr)
    padded = np.zeros((4, slots.shape[1], slots.shape[2]))
    padded[:slots.shape[0]] = slots                       # raises for > 4 examples, like the reference
    return np.concatenate(np.swapaxes(padded, 1, 2)[:4], axis=1)


def overlapping_split(spec, num_frames, frame_length):
    """dataset.py:352-357 -- windows at stride (width - frame_length) // (num_frames - 1)."""
    stride = (spec.shape[1] - frame_length) // (num_frames - 1)
    return np.array([spec[:, i:i + frame_length] for i in range(0, spec.shape[1], stride)][:num_frames])


def contiguous_split(spec, num_frames, frame_length):
    """dataset.py:360-361 -- back-to-back windows (at most width // frame_length of them)."""
    return np.array([spec[:, i:i + frame_length] for i in range(0, spec.shape[1], frame_length)][:num_frames])


def split(spec, num_frames, x_size, y_size, overlap):
    frames = overlapping_split(spec, num_frames, x_size) if overlap else contiguous_split(spec, num_frames, x_size)
    assert all(f.shape == (y_size, x_size) for f in frames)
    return frames
