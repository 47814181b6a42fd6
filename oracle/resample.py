"""CPU restatement (numpy float64) of the resampling branch of the reference, vggish_input.py:52-53:
``data = resampy.resample(data, sample_rate, vggish_params.SAMPLE_RATE)`` with resampy's default filter 'kaiser_best'.

TEST INFRASTRUCTURE ONLY (imported by tests/, never by the product path).

PARITY UNPINNED. resampy (a third-party dependency of the reference; requirements.txt lists it without a version) is not
installed in this environment, the reference tree holds no fixture of a resampled waveform, and there is no network: this
file restates resampy's PUBLISHED algorithm -- J. O. Smith's band-limited sinc interpolation ("Digital Audio Resampling Home
Page", the method resampy's documentation cites), as implemented in resampy/interpn.py -- and its published 'kaiser_best'
filter design (resampy/filters.py sinc_window with num_zeros = 64, precision = 9 i.e. 512 table entries per zero crossing,
Kaiser window beta = 14.769656459379492, rolloff = 0.9475937167399596; resampy ships that table as a data file).
The HIP kernel is checked against this restatement; neither is checked against resampy itself.
"""

import numpy as np

NUM_ZEROS = 64
PRECISION = 9
BETA = 14.769656459379492
ROLLOFF = 0.9475937167399596


def sinc_window(num_zeros=NUM_ZEROS, precision=PRECISION, beta=BETA, rolloff=ROLLOFF):
    """resampy/filters.py sinc_window: the right half of a Kaiser-windowed sinc low-pass, `2**precision` entries per zero crossing.
    Returns (interp_win float64 (num_zeros * 2**precision + 1,), num_table = 2**precision)."""
    num_bits = 2 ** precision
    n = num_bits * num_zeros
    sinc_win = rolloff * np.sinc(rolloff * np.linspace(0, num_zeros, num=n + 1, endpoint=True))
    taper = np.kaiser(2 * n + 1, beta)[n:]
    return taper * sinc_win, num_bits


def resample(x, sr_orig, sr_new):
    """resampy.resample(x, sr_orig, sr_new, filter='kaiser_best') for a 1-D signal (resampy/core.py + interpn.py resample_f):
    output length int(len(x) * sr_new / sr_orig); each output sample is the inner product of the input with the windowed sinc
    centred on its (fractional) input position, the filter read from the table with linear interpolation between entries."""
    x = np.asarray(x, dtype=np.float64)
    assert x.ndim == 1
    if sr_orig <= 0 or sr_new <= 0:
        raise ValueError("Invalid sample rate")
    ratio = float(sr_new) / float(sr_orig)
    n_out = int(x.shape[0] * ratio)
    if n_out < 1:
        raise ValueError("Input signal length=%d is too small to resample from %s->%s" % (x.shape[0], sr_orig, sr_new))
    win, num_table = sinc_window()
    if ratio < 1:
        win = win * ratio
    delta = np.zeros_like(win)
    delta[:-1] = np.diff(win)
    scale = min(1.0, ratio)
    index_step = int(scale * num_table)
    nwin = win.shape[0]
    n_orig = x.shape[0]
    t_reg = np.arange(n_out, dtype=np.float64) * (1.0 / ratio)           # position of every output sample on the input time axis
    n = t_reg.astype(np.int64)
    y = np.zeros(n_out, dtype=np.float64)
    # left wing: x[n - i], i = 0 .. i_max - 1
    frac = scale * (t_reg - n)
    index_frac = frac * num_table
    offset = index_frac.astype(np.int64)
    eta = index_frac - offset
    i_max = np.minimum(n + 1, (nwin - offset) // index_step)
    for i in range(int(i_max.max())):
        m = i < i_max
        idx = offset[m] + i * index_step
        y[m] += (win[idx] + eta[m] * delta[idx]) * x[n[m] - i]
    # right wing: x[n + k + 1], k = 0 .. k_max - 1
    frac = scale - frac
    index_frac = frac * num_table
    offset = index_frac.astype(np.int64)
    eta = index_frac - offset
    k_max = np.minimum(n_orig - n - 1, (nwin - offset) // index_step)
    for k in range(int(k_max.max())):
        m = k < k_max
        idx = offset[m] + k * index_step
        y[m] += (win[idx] + eta[m] * delta[idx]) * x[n[m] + k + 1]
    return y
