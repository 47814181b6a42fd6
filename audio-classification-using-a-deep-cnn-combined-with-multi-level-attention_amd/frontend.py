"""Host side of the HIP audio front-end: device tables, buffer plumbing, ctypes calls.

PyTorch is used for device memory and streams only; all arithmetic is in csrc/logmel.hip.
"""

import ctypes

import numpy as np
import torch

from . import _lib

_tables = {}


def device_tables(device):
    """Constant tables of the fused kernel, built by the library in float64 and uploaded once."""
    key = str(device)
    if key not in _tables:
        L = _lib.lib()
        n = int(L.mla_logmel_table_floats())
        host = np.zeros(n, dtype=np.float32)
        _lib.check(L.mla_logmel_build_tables(host.ctypes.data_as(ctypes.c_void_p)))
        _tables[key] = torch.from_numpy(host).to(device)
    return _tables[key]


def counts(n_samples):
    """(stft_frames, examples) for a 16 kHz waveform; ValueError where the reference raises."""
    L = _lib.lib()
    f, e = ctypes.c_int64(), ctypes.c_int64()
    rc = L.mla_logmel_counts(int(n_samples), ctypes.byref(f), ctypes.byref(e))
    if rc == _lib.E_SHORT:
        raise ValueError("negative dimensions are not allowed")
    _lib.check(rc)
    return f.value, e.value


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("the HIP front-end needs a GPU (cuda:0); there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def as_device_mono(data, pcm16=False):
    """ndarray / tensor, 1-D or (samples, channels) -> 1-D device tensor.

    pcm16=False (the public ``waveform_to_examples``, vggish_input.py:30-82, which never scales its input): integer
    arrays are converted to float32 VALUES as they are, like numpy's promotion in the reference.
    pcm16=True (``wavfile_to_examples`` and the PCM streaming paths, vggish_input.py:97-98 ``wav_data / 32768.0``): mono
    int16 stays int16 and the log-mel kernel scales by 1/32768 in its PCM read; multi-channel int16 is scaled in the mix.
    Multi-channel input is averaged over axis 1 as vggish_input.py:49-50 does, by `mla_mono_mix` on the device
    (double-precision mean, one rounding).
    """
    dev = _device()
    if isinstance(data, np.ndarray):
        if data.dtype != np.int16 or not pcm16:
            data = data.astype(np.float32, copy=False)     # float64 reference input: rounded once, like the mono path
        t = torch.from_numpy(np.array(data, order="C", copy=True) if not data.flags.writeable else np.ascontiguousarray(data)).to(dev)
    else:
        t = data.to(dev)
        if t.dtype != torch.int16 or not pcm16:
            t = t.float()
        t = t.contiguous()
    if t.dim() == 1:
        return t
    assert t.dim() == 2, "waveform must be 1-D or (samples, channels)"
    n, ch = t.shape
    out = torch.empty((n,), dtype=torch.float32, device=dev)
    code = _lib.I16 if t.dtype == torch.int16 else _lib.F32
    _lib.check(_lib.lib().mla_mono_mix(ctypes.c_void_p(t.data_ptr()), code, n, ch, ctypes.c_void_p(out.data_ptr()), _lib.stream_ptr()))
    return out


_resample_tables = {}


def kaiser_best_filter():
    """resampy's default interpolation filter 'kaiser_best' (resampy/filters.py sinc_window: 64 zero crossings, 2**9 table entries
    per crossing, Kaiser beta 14.769656459379492, rolloff 0.9475937167399596), right half, float64. resampy ships this table as a
    data file; it is host-side setup like the mel matrix. (interp_win, num_table)."""
    num_zeros, precision, beta, rolloff = 64, 9, 14.769656459379492, 0.9475937167399596
    num_bits = 2 ** precision
    n = num_bits * num_zeros
    sinc_win = rolloff * np.sinc(rolloff * np.linspace(0, num_zeros, num=n + 1, endpoint=True))
    return np.kaiser(2 * n + 1, beta)[n:] * sinc_win, num_bits


def resample(wave, sr_orig, sr_new, gain=1.0):
    """resampy.resample(wave, sr_orig, sr_new) (vggish_input.py:52-53) for a 1-D device waveform -> float32 device tensor of
    int(n * sr_new / sr_orig) samples, by the HIP kernel mla_resample (`gain` scales the filter table: the operation is linear).
    ValueError where resampy raises (bad rates, empty result)."""
    if sr_orig <= 0:
        raise ValueError("Invalid sample rate: sr_orig=%r" % (sr_orig,))
    if sr_new <= 0:
        raise ValueError("Invalid sample rate: sr_new=%r" % (sr_new,))
    assert wave.dim() == 1 and wave.is_cuda
    wave = wave.float().contiguous()
    ratio = float(sr_new) / float(sr_orig)
    L = _lib.lib()
    n_out = int(L.mla_resample_length(wave.shape[0], float(sr_orig), float(sr_new)))
    if n_out < 1:
        raise ValueError("Input signal length=%d is too small to resample from %s->%s" % (wave.shape[0], sr_orig, sr_new))
    key = (str(wave.device), ratio if ratio < 1 else 1.0, float(gain))
    if key not in _resample_tables:
        win, num_table = kaiser_best_filter()
        if ratio < 1:
            win = win * ratio
        win = win * float(gain)
        delta = np.zeros_like(win)
        delta[:-1] = np.diff(win)
        _resample_tables[key] = (torch.from_numpy(win).to(wave.device), torch.from_numpy(delta).to(wave.device), num_table)
    win, delta, num_table = _resample_tables[key]
    out = torch.empty(n_out, dtype=torch.float32, device=wave.device)
    vp = ctypes.c_void_p
    _lib.check(L.mla_resample(vp(wave.data_ptr()), wave.shape[0], float(sr_orig), float(sr_new), vp(win.data_ptr()), vp(delta.data_ptr()),
                              win.shape[0], num_table, vp(out.data_ptr()), n_out, _lib.stream_ptr()))
    return out


def waveforms_to_examples(pcm, out_dtype=torch.float32, out=None):
    """(W, n) device PCM (float32 or int16) -> (W * N, 96, 64) examples, waveform-major."""
    assert pcm.dim() == 2 and pcm.is_cuda and pcm.stride(1) == 1
    n_wave, n_samples = pcm.shape
    _, n_ex = counts(n_samples)
    if out is None:
        out = torch.empty((n_wave * n_ex, 96, 64), dtype=out_dtype, device=pcm.device)
    else:
        assert out.is_contiguous() and out.numel() == n_wave * n_ex * 96 * 64 and out.dtype == out_dtype
    if n_wave * n_ex == 0:
        return out
    pcm_code = {torch.float32: _lib.F32, torch.int16: _lib.I16}[pcm.dtype]
    out_code = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16}[out_dtype]
    tab = device_tables(pcm.device)
    from . import ops
    _lib.check(ops._timed("logmel", _lib.lib().mla_logmel_examples,
                          ctypes.c_void_p(pcm.data_ptr()), pcm_code, n_wave, n_samples, pcm.stride(0) if n_wave > 1 else n_samples,   # (a size-1 dimension's stride is arbitrary)
                          ctypes.c_void_p(tab.data_ptr()), ctypes.c_void_p(out.data_ptr()), out_code, _lib.stream_ptr()))
    return out


def _as_device_signal(signal):
    dev = _device()
    if isinstance(signal, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(signal.astype(np.float32))).to(dev)
    return signal.to(dev).float().contiguous()


def stft_magnitude(signal, fft_length, hop_length, window_length):
    """mel_features.stft_magnitude (mel_features.py:71-92) on the GPU for any window / hop and
    power-of-two fft_length <= 4096: (frames, fft_length/2 + 1) float32 CUDA tensor."""
    from .torchvggish import mel_features
    x = _as_device_signal(signal)
    assert x.dim() == 1
    n = x.shape[0]
    frames = 1 + int(np.floor((n - window_length) / hop_length))
    if frames < 0:
        raise ValueError("negative dimensions are not allowed")
    bins = fft_length // 2 + 1
    out = torch.empty((frames, bins), dtype=torch.float32, device=x.device)
    if frames == 0:
        return out
    window = torch.from_numpy(mel_features.periodic_hann(window_length).astype(np.float32)).to(x.device)
    m = np.arange(fft_length // 2)
    tw = np.stack([np.cos(2 * np.pi * m / fft_length), -np.sin(2 * np.pi * m / fft_length)], axis=1).astype(np.float32)
    tw = torch.from_numpy(np.ascontiguousarray(tw)).to(x.device)
    vp = ctypes.c_void_p
    _lib.check(_lib.lib().mla_stft_magnitude(vp(x.data_ptr()), n, vp(window.data_ptr()), vp(tw.data_ptr()), int(window_length),
                                             int(hop_length), int(fft_length), vp(out.data_ptr()), _lib.stream_ptr()))
    return out


def log_mel_spectrogram(data, audio_sample_rate, log_offset, window_length_secs, hop_length_secs, **kwargs):
    """mel_features.log_mel_spectrogram (mel_features.py:192-223) on the GPU for any configuration:
    (frames, num_mel_bins) float32 CUDA tensor. The mel matrix is host-side setup, as in the reference."""
    from .torchvggish import mel_features
    win = int(round(audio_sample_rate * window_length_secs))
    hop = int(round(audio_sample_rate * hop_length_secs))
    fft = 2 ** int(np.ceil(np.log(win) / np.log(2.0)))
    spec = stft_magnitude(data, fft, hop, win)
    mel = mel_features.spectrogram_to_mel_matrix(num_spectrogram_bins=spec.shape[1], audio_sample_rate=audio_sample_rate, **kwargs)
    melt = torch.from_numpy(np.ascontiguousarray(mel.astype(np.float32))).to(spec.device)
    out = torch.empty((spec.shape[0], mel.shape[1]), dtype=torch.float32, device=spec.device)
    vp = ctypes.c_void_p
    _lib.check(_lib.lib().mla_mel_log(vp(spec.data_ptr()), vp(melt.data_ptr()), spec.shape[0], spec.shape[1], mel.shape[1],
                                      float(log_offset), vp(out.data_ptr()), _lib.stream_ptr()))
    return out
