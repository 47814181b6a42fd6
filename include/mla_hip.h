/*
 * mla_hip.h -- C ABI of libmla_hip.so: the MI355X (gfx950) hot path of
 * caesar-one/audio-classification-using-a-deep-cnn-combined-with-multi-level-attention
 *
 *   waveform -> framed STFT -> mel -> log -> VGGish conv stack -> FC embeddings
 *            -> multi-level attention pooling -> class scores  (+ the training step)
 *
 * The reference has no FFI: its callers use Python functions / nn.Module classes
 * directly. Each entry point below names the reference
 * interface (file:line under the reference tree) whose arithmetic it replaces; the
 * Python drop-in modules in the package bind them with ctypes.
 *
 * Conventions
 *   - plain C: raw DEVICE pointers (unless a parameter says "host"), explicit int64
 *     sizes, a hipStream_t passed as void*, caller-provided workspaces;
 *   - every function returns 0 on success or a negative MLA_E_* code; the message is
 *     available from mla_last_error() (thread-local);
 *   - no allocation, no host synchronisation and no implicit stream inside: kernels are
 *     enqueued on the given stream and the caller keeps the buffers alive until they
 *     have run (graph-capturable);
 *   - shapes are checked on the host BEFORE any launch; a shape the kernels were not
 *     compiled for is an error, never a silent fallback.
 */
#ifndef MLA_HIP_H
#define MLA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MLA_OK            0
#define MLA_E_ARG        -1   /* null pointer / negative size / misaligned buffer        */
#define MLA_E_SHAPE      -2   /* shape not supported by the compiled kernels             */
#define MLA_E_SHORT      -3   /* waveform shorter than 240 samples (reference: ValueError,
                                 mel_features.py:42-45 via vggish_input.py:56)           */
#define MLA_E_LAUNCH     -4   /* hipLaunch / HIP runtime error                           */
#define MLA_E_DTYPE      -5   /* unknown dtype code                                      */

/* element types of activations / weights */
#define MLA_F32   0
#define MLA_BF16  1
#define MLA_I16   2

typedef void* mla_stream_t;            /* hipStream_t */

int         mla_abi_version(void);
const char* mla_last_error(void);

/* ------------------------------------------------------------------------------------
 * Front-end: torchvggish/mel_features.py + torchvggish/vggish_input.py
 * ---------------------------------------------------------------------------------- */

/* mel_features.py:42 and vggish_input.py:67-76 -- exact integer frame arithmetic for a
 * 16 kHz waveform: stft_frames = 1 + floor((n-400)/160), examples = 1 + floor((F-96)/96)
 * clamped at 0. Returns MLA_E_SHORT when the reference would raise (n < 240). Host only. */
int mla_logmel_counts(int64_t n_samples, int64_t* stft_frames, int64_t* examples);

/* Constant tables of the fused kernel (periodic Hann mel_features.py:48-68, HTK mel
 * matrix mel_features.py:114-189 in sparse per-lane form, FFT twiddles), computed in
 * double precision on the host. `host_out` receives mla_logmel_table_floats() floats;
 * the caller uploads them once and passes the device copy to mla_logmel_examples. */
int64_t mla_logmel_table_floats(void);
int     mla_logmel_build_tables(float* host_out);
/* Dense (257 x 64) double-precision mel matrix and 400-point window exactly as the
 * tables encode them (host; used by the CPU tests to pin the tables to the oracle). */
int     mla_logmel_reference_tables(double* host_window400, double* host_mel_257x64);

/* vggish_input.waveform_to_examples (vggish_input.py:30-82) for 16 kHz mono input:
 * pcm[n_wave][wave_stride] (first n_samples of each row used) -> examples
 * out[n_wave * examples][96][64], waveform-major, where examples comes from
 * mla_logmel_counts(n_samples). pcm_dtype: MLA_F32 (samples in [-1,1)) or MLA_I16
 * (scaled by 1/32768 as vggish_input.py:98 does). out_dtype: MLA_F32 or MLA_BF16.
 * One fused kernel: frame (mel_features.py:21-45) -> Hann -> |rfft 512|
 * (mel_features.py:71-92) -> mel (mel_features.py:220) -> log(. + 0.01) (:223) ->
 * 96-frame examples (vggish_input.py:73-76). */
int mla_logmel_examples(const void* pcm, int pcm_dtype, int64_t n_wave, int64_t n_samples,
                        int64_t wave_stride, const float* tables, void* out, int out_dtype,
                        mla_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MLA_HIP_H */
