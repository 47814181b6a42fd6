"""Drop-in for the reference's ``torchvggish/vggish_input.py`` (vggish_input.py:30, :85).

``waveform_to_examples`` keeps the reference's signature and semantics (mono mix over axis
1, 16 kHz assumed, (N, 96, 64) examples, optional ``(N, 1, 96, 64)`` float32 tensor with
``requires_grad=True``) but runs as ONE fused HIP kernel and returns CUDA tensors; nothing
on this path is computed by numpy or torch ops.
"""

import numpy as np
import torch

from . import vggish_params
from .. import frontend


def waveform_to_examples(data, sample_rate, return_tensor=True, _pcm16=False):
    """Waveform -> VGGish examples.

    data: np.ndarray or torch tensor, 1-D mono or 2-D ``(samples, channels)`` (averaged over
    axis 1, vggish_input.py:49-50). Values are used as they are, whatever the dtype: like the
    reference this function never rescales (only ``wavfile_to_examples`` divides int16 PCM by 32768).
    Returns ``(N, 1, 96, 64)`` float32 CUDA tensor (requires_grad=True, vggish_input.py:79-80)
    or, with ``return_tensor=False``, an ``(N, 96, 64)`` float64 ndarray like the reference.
    """
    if sample_rate != vggish_params.SAMPLE_RATE:
        # vggish_input.py:52-53: resampy.resample(data, sample_rate, 16000) -- band-limited sinc interpolation with resampy's
        # 'kaiser_best' filter, as a HIP kernel. resampy is not installed here, so parity with it is unpinned.
        # Resampling is linear: wavfile_to_examples' 1/32768 (vggish_input.py:98) rides in the filter table.
        mono = frontend.resample(frontend.as_device_mono(data), sample_rate, vggish_params.SAMPLE_RATE,
                                 gain=1.0 / 32768.0 if _pcm16 else 1.0)
    else:
        mono = frontend.as_device_mono(data, pcm16=_pcm16)
    examples = frontend.waveforms_to_examples(mono[None], out_dtype=torch.float32)
    if return_tensor:
        return examples[:, None, :, :].requires_grad_(True)
    return examples.cpu().numpy().astype(np.float64)


def wavfile_to_examples(wav_file, return_tensor=True):
    """WAV file (signed 16-bit PCM) -> examples (vggish_input.py:85-99). The int16 -> [-1, 1)
    scaling (/32768) is fused into the kernel's PCM read."""
    import wave
    with wave.open(wav_file, "rb") as wf:
        assert wf.getsampwidth() == 2, "Bad sample type: %r" % wf.getsampwidth()
        sr, ch = wf.getframerate(), wf.getnchannels()
        pcm = np.frombuffer(wf.readframes(wf.getnframes()), dtype=np.int16)
    if ch > 1:
        pcm = pcm.reshape(-1, ch)
    return waveform_to_examples(pcm, sr, return_tensor, _pcm16=True)
