"""Drop-in for the reference's ``torchvggish/mel_features.py`` (same names, argument order
and defaults: mel_features.py:21, :48, :71, :100, :114, :192), backed by the HIP front-end.

Constant tables (Hann window, mel matrix) are host-side setup and are computed in float64
exactly as the reference does; the per-sample arithmetic (framing, window, FFT, magnitude,
mel, log) runs in ``csrc/logmel.hip`` through ``libmla_hip.so`` -- there is no numpy
fallback for it. ``frame`` stays what it is in the reference: a zero-copy strided view.
"""

import numpy as np

_MEL_BREAK_FREQUENCY_HERTZ = 700.0
_MEL_HIGH_FREQUENCY_Q = 1127.0


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def frame(data, window_length, hop_length):
    """(num_samples, ...) -> view (num_frames, window_length, ...), frame i starting at
    i * hop_length; incomplete trailing frames are dropped, nothing is copied
    (mel_features.py:21-45). Works on numpy arrays and on torch tensors (any device)."""
    num_samples = data.shape[0]
    num_frames = 1 + int(np.floor((num_samples - window_length) / hop_length))
    if num_frames < 0:
        raise ValueError("negative dimensions are not allowed")
    shape = (num_frames, window_length) + tuple(data.shape[1:])
    if _is_torch(data):
        strides = (data.stride(0) * hop_length,) + tuple(data.stride())
        return data.as_strided(shape, strides)
    strides = (data.strides[0] * hop_length,) + data.strides
    return np.lib.stride_tricks.as_strided(data, shape=shape, strides=strides)


def periodic_hann(window_length):
    """Raised cosine over one full period of window_length samples (mel_features.py:48-68)."""
    phase = 2 * np.pi / window_length * np.arange(window_length)
    return 0.5 - 0.5 * np.cos(phase)


def hertz_to_mel(frequencies_hertz):
    """HTK mel scale (mel_features.py:100-111)."""
    return _MEL_HIGH_FREQUENCY_Q * np.log(1.0 + (frequencies_hertz / _MEL_BREAK_FREQUENCY_HERTZ))


def spectrogram_to_mel_matrix(num_mel_bins=20, num_spectrogram_bins=129, audio_sample_rate=8000,
                              lower_edge_hertz=125.0, upper_edge_hertz=3800.0):
    """(num_spectrogram_bins, num_mel_bins) triangular filterbank, linear in mel, DC row zero;
    raises ValueError on bad edges like mel_features.py:156-163."""
    nyquist_hertz = audio_sample_rate / 2.0
    if lower_edge_hertz < 0.0:
        raise ValueError("lower_edge_hertz %.1f must be >= 0" % lower_edge_hertz)
    if lower_edge_hertz >= upper_edge_hertz:
        raise ValueError("lower_edge_hertz %.1f >= upper_edge_hertz %.1f" % (lower_edge_hertz, upper_edge_hertz))
    if upper_edge_hertz > nyquist_hertz:
        raise ValueError("upper_edge_hertz %.1f is greater than Nyquist %.1f" % (upper_edge_hertz, nyquist_hertz))
    bin_mel = hertz_to_mel(np.linspace(0.0, nyquist_hertz, num_spectrogram_bins))[:, np.newaxis]
    edges = np.linspace(hertz_to_mel(lower_edge_hertz), hertz_to_mel(upper_edge_hertz), num_mel_bins + 2)
    lower, center, upper = edges[np.newaxis, :-2], edges[np.newaxis, 1:-1], edges[np.newaxis, 2:]
    weights = np.maximum(0.0, np.minimum((bin_mel - lower) / (center - lower), (upper - bin_mel) / (upper - center)))
    weights[0, :] = 0.0
    return weights


def stft_magnitude(signal, fft_length, hop_length=None, window_length=None):
    """|rfft(frame * periodic_hann, fft_length)| -> (num_frames, fft_length/2 + 1)
    (mel_features.py:71-92), computed on the GPU. Returns a float32 CUDA tensor."""
    from .. import frontend
    return frontend.stft_magnitude(signal, int(fft_length), hop_length, window_length)


def log_mel_spectrogram(data, audio_sample_rate=8000, log_offset=0.0, window_length_secs=0.025,
                        hop_length_secs=0.010, **kwargs):
    """log(|STFT| . mel + log_offset) -> (num_frames, num_mel_bins) (mel_features.py:192-223),
    computed on the GPU. Returns a float32 CUDA tensor."""
    from .. import frontend
    return frontend.log_mel_spectrogram(data, audio_sample_rate, log_offset, window_length_secs,
                                        hop_length_secs, **kwargs)
