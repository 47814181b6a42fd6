"""ORACLE (test infrastructure, not product code): numpy float64 restatement of the
reference's audio front-end.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module; the product path (the HIP kernels behind
``include/mla_hip.h``) never routes through it.

Pinned: ``tests/test_oracle_golden.py`` checks every function below against
``tests/golden/frontend_*.npz``, which ``tests/golden/make_golden.py`` produced by
importing the reference itself (``/root/reference/torchvggish``) in the build
container, plus the known-answer constants of SURVEY.md section 4.

Each function cites the reference lines it restates (paths relative to
``/root/reference``).
"""

import math

import numpy as np

SAMPLE_RATE = 16000          # vggish_params.py:27
WINDOW_SEC, HOP_SEC = 0.025, 0.010   # vggish_params.py:28-29
NUM_MEL = 64                 # vggish_params.py:30
MEL_LO_HZ, MEL_HI_HZ = 125.0, 7500.0  # vggish_params.py:31-32
LOG_OFFSET = 0.01            # vggish_params.py:33
EX_FRAMES = 96               # vggish_params.py:22, :34-35 (0.96 s window == hop)


def num_frames(num_samples, window_length, hop_length):
    """mel_features.py:42 -- complete frames only, no padding (may be <= 0)."""
    return 1 + int(math.floor((num_samples - window_length) / hop_length))


def frame(data, window_length, hop_length):
    """mel_features.py:21-45 -- out[i, j, ...] = data[i*hop + j, ...].

    The reference returns a strided view; a gathered copy holds the same values.
    A negative frame count raises ValueError like numpy's as_strided does
    ("negative dimensions are not allowed").
    """
    data = np.asarray(data)
    nf = num_frames(data.shape[0], window_length, hop_length)
    if nf < 0:
        raise ValueError("negative dimensions are not allowed")
    idx = (np.arange(nf)[:, None] * hop_length + np.arange(window_length)[None, :])
    return data[idx.reshape(-1)].reshape((nf, window_length) + data.shape[1:])


def periodic_hann(window_length):
    """mel_features.py:48-68 -- 0.5 - 0.5 cos(2 pi n / N), a full period-N cosine."""
    n = np.arange(window_length)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi / window_length * n)


def stft_magnitude(signal, fft_length, hop_length, window_length):
    """mel_features.py:71-92 -- |rfft(frame * hann, n=fft_length)|, bins 0..fft/2."""
    frames = frame(np.asarray(signal, dtype=np.float64), window_length, hop_length)
    return np.abs(np.fft.rfft(frames * periodic_hann(window_length), int(fft_length)))


def hertz_to_mel(hz):
    """mel_features.py:100-111 -- HTK mel: 1127 ln(1 + f/700)."""
    return 1127.0 * np.log(1.0 + np.asarray(hz, dtype=np.float64) / 700.0)


def mel_matrix(num_mel_bins=20, num_spectrogram_bins=129, audio_sample_rate=8000,
               lower_edge_hertz=125.0, upper_edge_hertz=3800.0):
    """mel_features.py:114-189 -- (bins, bands) triangular weights, linear in mel,
    DC row zeroed (:188); same three ValueErrors as :156-163."""
    nyquist = audio_sample_rate / 2.0
    if lower_edge_hertz < 0.0:
        raise ValueError("lower_edge_hertz %.1f must be >= 0" % lower_edge_hertz)
    if lower_edge_hertz >= upper_edge_hertz:
        raise ValueError("lower_edge_hertz %.1f >= upper_edge_hertz %.1f"
                         % (lower_edge_hertz, upper_edge_hertz))
    if upper_edge_hertz > nyquist:
        raise ValueError("upper_edge_hertz %.1f is greater than Nyquist %.1f"
                         % (upper_edge_hertz, nyquist))
    bins_mel = hertz_to_mel(np.linspace(0.0, nyquist, num_spectrogram_bins))
    edges = np.linspace(hertz_to_mel(lower_edge_hertz), hertz_to_mel(upper_edge_hertz),
                        num_mel_bins + 2)
    lo, ce, hi = edges[:-2], edges[1:-1], edges[2:]
    rising = (bins_mel[:, None] - lo[None, :]) / (ce - lo)[None, :]
    falling = (hi[None, :] - bins_mel[:, None]) / (hi - ce)[None, :]
    w = np.maximum(0.0, np.minimum(rising, falling))
    w[0, :] = 0.0
    return w


def stft_config(audio_sample_rate, window_length_secs, hop_length_secs):
    """mel_features.py:212-214 -- (window, hop, fft) in samples."""
    win = int(round(audio_sample_rate * window_length_secs))
    hop = int(round(audio_sample_rate * hop_length_secs))
    fft = 2 ** int(np.ceil(np.log(win) / np.log(2.0)))
    return win, hop, fft


def log_mel_spectrogram(data, audio_sample_rate=8000, log_offset=0.0,
                        window_length_secs=0.025, hop_length_secs=0.010, **mel_kwargs):
    """mel_features.py:192-223 -- log(|STFT| @ mel + offset), natural log, float64."""
    win, hop, fft = stft_config(audio_sample_rate, window_length_secs, hop_length_secs)
    spec = stft_magnitude(data, fft, hop, win)
    mel = spec @ mel_matrix(num_spectrogram_bins=spec.shape[1],
                            audio_sample_rate=audio_sample_rate, **mel_kwargs)
    return np.log(mel + log_offset)


def waveform_to_examples(data, sample_rate=SAMPLE_RATE):
    """vggish_input.py:30-82 with return_tensor=False -- (N, 96, 64) float64.

    Mono mix is mean over axis 1 (:49-50). Resampling (:52-53, resampy) is outside
    the hot path: the oracle only accepts 16 kHz.
    """
    data = np.asarray(data, dtype=np.float64)
    if data.ndim > 1:
        data = data.mean(axis=1)
    if sample_rate != SAMPLE_RATE:
        raise NotImplementedError("oracle covers the 16 kHz path only (vggish_input.py:52-53 is off-path)")
    log_mel = log_mel_spectrogram(data, audio_sample_rate=SAMPLE_RATE, log_offset=LOG_OFFSET,
                                  window_length_secs=WINDOW_SEC, hop_length_secs=HOP_SEC,
                                  num_mel_bins=NUM_MEL, lower_edge_hertz=MEL_LO_HZ,
                                  upper_edge_hertz=MEL_HI_HZ)
    return frame(log_mel, EX_FRAMES, EX_FRAMES)


def num_examples(num_samples):
    """Examples produced for a 16 kHz waveform of num_samples (0 if too short,
    ValueError below 240 samples as mel_features.py:42-45 implies)."""
    f = num_frames(num_samples, 400, 160)
    if f < 0:
        raise ValueError("negative dimensions are not allowed")
    return max(0, num_frames(f, EX_FRAMES, EX_FRAMES))


def batch_examples(waves):
    """(W, n) waveforms -> (W * N, 96, 64) float64, waveform-major."""
    return np.concatenate([waveform_to_examples(w) for w in np.asarray(waves)], axis=0)
