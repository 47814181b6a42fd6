// stft_generic.hip -- stand-alone mel_features.stft_magnitude / log_mel_spectrogram for ARBITRARY
// configurations (reference API: mel_features.py:71-92 and :192-223 accept any window / hop /
// power-of-two FFT length / mel layout). The VGGish configuration on the hot path is served by the
// fused kernel in logmel.hip; these kernels exist so that the drop-in module keeps the reference's
// full argument space on the GPU. They are simple (one workgroup per frame, radix-2 FFT in LDS,
// dense mel product) and not tuned.
#include "common.h"

namespace {

constexpr int kMaxFft = 4096;

// One workgroup per STFT frame: x[frame*hop + n] * window[n] (n < win), zero-padded to `fft`,
// in-place iterative radix-2 decimation-in-time FFT on (re, im) in LDS, then |X[k]|, k <= fft/2.
// twiddle: fft/2 pairs (cos, -sin)(2 pi m / fft) computed on the host in double precision.
__global__ __launch_bounds__(256) void stft_kernel(const float* __restrict__ signal, const float* __restrict__ window,
                                                   const float* __restrict__ twiddle, int win, int hop, int fft, int log2fft,
                                                   float* __restrict__ out) {
    extern __shared__ float lds[];
    float* re = lds;
    float* im = lds + fft;
    const int64_t frame = blockIdx.x;
    const float* x = signal + frame * hop;
    for (int n = threadIdx.x; n < fft; n += 256) {
        const unsigned rev = __brev(unsigned(n)) >> (32 - log2fft);       // bit-reversed load position
        re[rev] = n < win ? x[n] * window[n] : 0.f;
        im[rev] = 0.f;
    }
    __syncthreads();
    for (int s = 1; s <= log2fft; ++s) {
        const int half = 1 << (s - 1), step = fft >> s;                    // twiddle index stride
        for (int b = threadIdx.x; b < fft / 2; b += 256) {
            const int grp = b / half, pos = b % half;
            const int i0 = grp * 2 * half + pos, i1 = i0 + half;
            const float wr = twiddle[2 * (pos * step)], wi = twiddle[2 * (pos * step) + 1];
            const float tr = re[i1] * wr - im[i1] * wi, ti = re[i1] * wi + im[i1] * wr;
            const float ur = re[i0], ui = im[i0];
            re[i0] = ur + tr; im[i0] = ui + ti;
            re[i1] = ur - tr; im[i1] = ui - ti;
        }
        __syncthreads();
    }
    const int bins = fft / 2 + 1;
    for (int k = threadIdx.x; k < bins; k += 256) out[frame * bins + k] = sqrtf(re[k] * re[k] + im[k] * im[k]);
}

// out[f][b] = log(sum_k spec[f][k] * mel[k][b] + offset)      (mel_features.py:220-223)
__global__ __launch_bounds__(256) void mel_log_kernel(const float* __restrict__ spec, const float* __restrict__ mel, int64_t frames,
                                                      int bins, int bands, float offset, float* __restrict__ out) {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= frames * bands) return;
    const int64_t f = i / bands;
    const int b = int(i - f * bands);
    float acc = 0.f;
    for (int k = 0; k < bins; ++k) acc = fmaf(spec[f * bins + k], mel[int64_t(k) * bands + b], acc);
    out[i] = logf(acc + offset);
}

}  // namespace

// mel_features.stft_magnitude for one 1-D signal: frames = 1 + floor((n - win)/hop) rows of fft/2+1 magnitudes.
extern "C" int mla_stft_magnitude(const float* signal, int64_t n_samples, const float* window, const float* twiddle,
                                  int64_t window_length, int64_t hop_length, int64_t fft_length, float* out,
                                  mla_stream_t stream) {
    MLA_REQUIRE(signal && window && twiddle && out, MLA_E_ARG, "null stft argument");
    MLA_REQUIRE(window_length >= 1 && hop_length >= 1 && fft_length >= window_length && fft_length <= kMaxFft &&
                    (fft_length & (fft_length - 1)) == 0 && fft_length >= 2,
                MLA_E_SHAPE, "stft needs a power-of-two fft_length in [max(2, window), %d] (got window %lld, fft %lld)", kMaxFft,
                (long long)window_length, (long long)fft_length);
    if (n_samples < window_length) return MLA_OK;          // zero frames (the caller sized `out` accordingly)
    const int64_t frames = 1 + (n_samples - window_length) / hop_length;
    MLA_REQUIRE(frames <= 0x7fffffff, MLA_E_SHAPE, "too many frames");
    int lg = 0;
    while ((1ll << lg) < fft_length) ++lg;
    hipLaunchKernelGGL(stft_kernel, dim3(unsigned(frames)), dim3(256), size_t(2 * fft_length * sizeof(float)),
                       static_cast<hipStream_t>(stream), signal, window, twiddle, int(window_length), int(hop_length),
                       int(fft_length), lg, out);
    MLA_LAUNCH_OK("stft_kernel");
    return MLA_OK;
}

extern "C" int mla_mel_log(const float* spectrogram, const float* mel_matrix, int64_t frames, int64_t bins, int64_t bands,
                           float log_offset, float* out, mla_stream_t stream) {
    MLA_REQUIRE(spectrogram && mel_matrix && out && frames >= 0 && bins >= 1 && bands >= 1, MLA_E_ARG, "bad mel_log arguments");
    if (frames == 0) return MLA_OK;
    hipLaunchKernelGGL(mel_log_kernel, dim3(unsigned((frames * bands + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), spectrogram, mel_matrix, frames, int(bins), int(bands), log_offset, out);
    MLA_LAUNCH_OK("mel_log_kernel");
    return MLA_OK;
}

// ---- dataset.create_spec (native path) + split, dataset.py:318-324 and :329-363 -------------------
// examples: (clips * ex_per_clip, 96, 64) from the front-end; out: (clips, n_frames, 64, frame_len):
//   out[c][t][band][x] = spec_c[band][t * stride + x],  spec_c[band][col] = ex[c][col / 96][col % 96][band]
// with the slots >= ex_per_clip of the 4-slot (64, 384) spectrogram zero (0.0, as the reference pads).
namespace {
__global__ __launch_bounds__(256) void dataset_frames_kernel(const float* __restrict__ ex, int ex_per_clip, int n_frames,
                                                             int frame_len, int stride, int64_t total, float* __restrict__ out) {
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < total; i += int64_t(gridDim.x) * 256) {
        const int x = int(i % frame_len);
        int64_t r = i / frame_len;
        const int band = int(r % 64); r /= 64;
        const int t = int(r % n_frames);
        const int64_t c = r / n_frames;
        const int col = t * stride + x, slot = col / 96, fr = col % 96;
        out[i] = (slot < ex_per_clip && col < 384) ? ex[((c * ex_per_clip + slot) * 96 + fr) * 64 + band] : 0.f;
    }
}
}  // namespace

extern "C" int mla_dataset_frames(const float* examples, int64_t clips, int ex_per_clip, int n_frames, int frame_len,
                                  int stride, float* out, mla_stream_t stream) {
    MLA_REQUIRE(out && clips >= 0 && ex_per_clip >= 0 && ex_per_clip <= 4 && n_frames >= 1 && frame_len >= 1 && stride >= 0,
                MLA_E_ARG, "bad dataset_frames arguments (a clip may hold at most 4 examples, dataset.py:321-322)");
    MLA_REQUIRE((n_frames - 1) * stride + frame_len <= 384, MLA_E_SHAPE, "frames leave the 384-column spectrogram");
    MLA_REQUIRE(examples || ex_per_clip == 0, MLA_E_ARG, "null examples");
    const int64_t total = clips * n_frames * 64 * frame_len;
    if (total == 0) return MLA_OK;
    hipLaunchKernelGGL(dataset_frames_kernel, dim3(unsigned((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), examples, ex_per_clip, n_frames, frame_len, stride, total, out);
    MLA_LAUNCH_OK("dataset_frames_kernel");
    return MLA_OK;
}

// ---- vggish.Postprocessor.postprocess, vggish.py:62-102 -------------------------------------------
// out[n][j] = round((clamp(sum_k E[j][k] (x[n][k] - mu[k]), -2, 2) + 2) * 63.75), round half to even
namespace {
__global__ __launch_bounds__(128) void postprocess_kernel(const float* __restrict__ x, const float* __restrict__ ev,
                                                          const float* __restrict__ mu, int64_t rows, float* __restrict__ out) {
    __shared__ float xc[128];
    const int64_t n = blockIdx.x;
    const int j = threadIdx.x;
    xc[j] = x[n * 128 + j] - mu[j];
    __syncthreads();
    float acc = 0.f;
    for (int k = 0; k < 128; ++k) acc = fmaf(ev[j * 128 + k], xc[k], acc);
    acc = fminf(fmaxf(acc, -2.0f), 2.0f);
    out[n * 128 + j] = rintf((acc + 2.0f) * (255.0f / 4.0f));
    (void)rows;
}
}  // namespace

template <typename T>
__global__ __launch_bounds__(256) void mono_mix_kernel(const T* __restrict__ pcm, int64_t n, int channels, double scale,
                                                        float* __restrict__ out) {
    const int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    if (i >= n) return;
    const T* p = pcm + i * channels;
    double acc = 0.0;                       // numpy's pairwise order is irrelevant: exact for int16, < 1 ulp of f64 for float
    for (int c = 0; c < channels; ++c) acc += double(p[c]);
    out[i] = float(acc / double(channels) * scale);
}

// ---- resampling to 16 kHz (vggish_input.py:52-53: resampy.resample(data, sample_rate, 16000), filter 'kaiser_best') ----
// Band-limited sinc interpolation (J. O. Smith) as resampy/interpn.py runs it: output sample t sits at input time
// t / ratio; it is the inner product of the input with the Kaiser-windowed sinc centred there, the filter read from a table of
// `num_table` entries per zero crossing with linear interpolation between entries (win + eta * delta), left wing over
// x[n], x[n-1], ..., right wing over x[n+1], x[n+2], ...; for ratio < 1 the table is stepped by int(ratio * num_table) entries per
// input sample (the time-stretched, gain-scaled low-pass). One lane per output sample; the ~2 * 64 / min(1, ratio) taps of a
// lane read consecutive input samples (L1/L2-resident: neighbouring lanes share all but one) and a table that fits L2 (512 KB
// in double precision). Double accumulation like the reference's float64 arrays; latency-bound, not a roofline kernel
// (a 10 s clip at 44.1 kHz is 160 000 outputs x 354 taps = 57 MFMA-free MFLOP).
__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ x, int64_t n_in, const double* __restrict__ win,
                                                       const double* __restrict__ delta, int nwin, int num_table, double ratio,
                                                       float* __restrict__ y, int64_t n_out) {
    const int64_t t = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (t >= n_out) return;
    const double scale = ratio < 1.0 ? ratio : 1.0;
    const int index_step = int(scale * num_table);
    const double treg = double(t) * (1.0 / ratio);
    const int64_t n = int64_t(treg);
    double acc = 0.0;
    double frac = scale * (treg - double(n));
    double index_frac = frac * num_table;
    int offset = int(index_frac);
    double eta = index_frac - offset;
    int64_t i_max = (nwin - offset) / index_step;
    if (n + 1 < i_max) i_max = n + 1;
    for (int64_t i = 0; i < i_max; ++i) {
        const int idx = offset + int(i) * index_step;
        acc += (win[idx] + eta * delta[idx]) * double(x[n - i]);
    }
    frac = scale - frac;
    index_frac = frac * num_table;
    offset = int(index_frac);
    eta = index_frac - offset;
    int64_t k_max = (nwin - offset) / index_step;
    if (n_in - n - 1 < k_max) k_max = n_in - n - 1;
    for (int64_t k = 0; k < k_max; ++k) {
        const int idx = offset + int(k) * index_step;
        acc += (win[idx] + eta * delta[idx]) * double(x[n + k + 1]);
    }
    y[t] = float(acc);
}

extern "C" int64_t mla_resample_length(int64_t n_in, double sr_in, double sr_out) {
    if (n_in < 0 || !(sr_in > 0.0) || !(sr_out > 0.0)) return -1;
    return int64_t(double(n_in) * (sr_out / sr_in));
}

extern "C" int mla_resample(const float* x, int64_t n_in, double sr_in, double sr_out, const double* win, const double* delta,
                            int nwin, int num_table, float* y, int64_t n_out, mla_stream_t stream) {
    MLA_REQUIRE(sr_in > 0.0 && sr_out > 0.0 && n_in >= 0 && nwin > 1 && num_table > 0, MLA_E_ARG, "bad resample arguments");
    MLA_REQUIRE(n_out == mla_resample_length(n_in, sr_in, sr_out), MLA_E_SHAPE, "n_out %lld != int(n_in * sr_out / sr_in) = %lld",
                (long long)n_out, (long long)mla_resample_length(n_in, sr_in, sr_out));
    MLA_REQUIRE(n_out >= 1, MLA_E_SHORT, "input of %lld samples is too short to resample from %g to %g Hz", (long long)n_in, sr_in, sr_out);
    const double ratio = sr_out / sr_in;
    MLA_REQUIRE(int((ratio < 1.0 ? ratio : 1.0) * num_table) >= 1, MLA_E_SHAPE, "ratio %g is below the filter table's resolution", ratio);
    MLA_REQUIRE(x && win && delta && y, MLA_E_ARG, "null resample buffers");
    hipLaunchKernelGGL(resample_kernel, dim3(unsigned((n_out + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x, n_in, win,
                       delta, nwin, num_table, ratio, y, n_out);
    MLA_LAUNCH_OK("resample_kernel");
    return MLA_OK;
}

extern "C" int mla_mono_mix(const void* pcm, int pcm_dtype, int64_t n_samples, int channels, float* out, mla_stream_t stream) {
    MLA_REQUIRE(n_samples >= 0 && channels >= 1, MLA_E_ARG, "bad n_samples %lld / channels %d", (long long)n_samples, channels);
    MLA_REQUIRE(pcm_dtype == MLA_F32 || pcm_dtype == MLA_I16, MLA_E_DTYPE, "pcm_dtype %d", pcm_dtype);
    if (n_samples == 0) return MLA_OK;
    MLA_REQUIRE(pcm && out, MLA_E_ARG, "null pcm/out");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const unsigned grid = unsigned((n_samples + 255) / 256);
    if (pcm_dtype == MLA_F32) {
        hipLaunchKernelGGL(mono_mix_kernel<float>, dim3(grid), dim3(256), 0, s, static_cast<const float*>(pcm), n_samples, channels, 1.0, out);
    } else {
        hipLaunchKernelGGL(mono_mix_kernel<int16_t>, dim3(grid), dim3(256), 0, s, static_cast<const int16_t*>(pcm), n_samples, channels,
                           1.0 / 32768.0, out);
    }
    MLA_LAUNCH_OK("mono_mix_kernel");
    return MLA_OK;
}

extern "C" int mla_postprocess(const float* embeddings, const float* pca_eigen_vectors, const float* pca_means, int64_t rows,
                               float* out, mla_stream_t stream) {
    MLA_REQUIRE(embeddings && pca_eigen_vectors && pca_means && out && rows >= 0, MLA_E_ARG, "bad postprocess arguments");
    if (rows == 0) return MLA_OK;
    hipLaunchKernelGGL(postprocess_kernel, dim3(unsigned(rows)), dim3(128), 0, static_cast<hipStream_t>(stream), embeddings,
                       pca_eigen_vectors, pca_means, rows, out);
    MLA_LAUNCH_OK("postprocess_kernel");
    return MLA_OK;
}
