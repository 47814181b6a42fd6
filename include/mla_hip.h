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
#define MLA_F64   4  /* mla_allreduce_flat only (the BatchNorm sums travel in double precision) */
#define MLA_I32   5  /* mla_allreduce_flat only (n_correct)                                        */
#define MLA_BF16X3 3  /* "split" bf16: x = hi + lo, two bf16 planes [hi(C) | lo(C)] per row / pixel; three bf16 MFMA
                        * products per term (hi*hi + hi*lo + lo*hi) with f32 accumulation: f32-grade results (2^-18
                        * relative per product) at a third of the bf16 rate. Accepted by mla_vggish_conv1 (output),
                        * mla_vggish_conv and mla_linear_bf16x3; weights are prepared by mla_split_bf16x3. */

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

/* Stand-alone stages for ARBITRARY configurations of the reference API (any window / hop /
 * power-of-two fft_length <= 4096, any mel layout); the VGGish configuration on the hot path
 * uses mla_logmel_examples instead.
 * mla_stft_magnitude: mel_features.stft_magnitude (mel_features.py:71-92) of one 1-D f32 signal;
 *   window: window_length floats (host-built periodic Hann); twiddle: fft_length/2 pairs
 *   (cos, -sin)(2 pi m / fft_length); out: frames x (fft_length/2 + 1) magnitudes,
 *   frames = 1 + floor((n_samples - window_length) / hop_length) (none if the signal is shorter).
 * mla_mel_log: log(spectrogram . mel_matrix + log_offset) (mel_features.py:220-223);
 *   spectrogram frames x bins, mel_matrix bins x bands (row-major), out frames x bands. */
int mla_stft_magnitude(const float* signal, int64_t n_samples, const float* window, const float* twiddle,
                       int64_t window_length, int64_t hop_length, int64_t fft_length, float* out,
                       mla_stream_t stream);
int mla_mel_log(const float* spectrogram, const float* mel_matrix, int64_t frames, int64_t bins, int64_t bands,
                float log_offset, float* out, mla_stream_t stream);

/* Mono mix of waveform_to_examples (vggish_input.py:49-50, `np.mean(data, axis=1)`), with wavfile_to_examples' int16
 * scaling (vggish_input.py:98) when the input is MLA_I16: interleaved (n_samples, channels) PCM -> (n_samples) float32.
 * The mean is taken in double precision like numpy's on the reference's float64 data, then rounded once. */
int mla_mono_mix(const void* pcm, int pcm_dtype, int64_t n_samples, int channels, float* out, mla_stream_t stream);
/* dataset.create_spec (native path, dataset.py:318-324) + split (dataset.py:329-363): the caller of
 * waveform_to_examples in the reference's data pipeline. examples (clips*ex_per_clip, 96, 64) ->
 * out (clips, n_frames, 64, frame_len) with out[c][t][band][x] = spec_c[band][t*stride + x], where
 * spec_c is the (64, 384) concatenation of the clip's <= 4 transposed examples, zero-padded. */
int mla_dataset_frames(const float* examples, int64_t clips, int ex_per_clip, int n_frames, int frame_len,
                       int stride, float* out, mla_stream_t stream);
/* vggish.Postprocessor.postprocess (vggish.py:62-102): PCA, clamp to [-2, 2], 8-bit quantisation
 * (as float). embeddings (rows, 128), pca_eigen_vectors (128, 128), pca_means (128). */
int mla_postprocess(const float* embeddings, const float* pca_eigen_vectors, const float* pca_means, int64_t rows,
                    float* out, mla_stream_t stream);

/* ------------------------------------------------------------------------------------
 * VGGish feature stack: torchvggish/vggish.py:108-118 (make_layers) applied at :22.
 * Activations are NHWC (N, H, W, C) in the compute dtype (MLA_BF16 or MLA_F32); this
 * makes the reference's NCHW->NHWC flatten (vggish.py:26-29, model.py:190-193) a no-op.
 * ---------------------------------------------------------------------------------- */

/* nn.Conv2d weight (Cout, Cin, 3, 3) f32 (state_dict layout) -> (Cout, 9, Cin) in `dtype`,
 * the K-contiguous layout the implicit-GEMM kernels stream. */
int mla_conv_repack_weights(const float* w_oihw, int64_t cout, int64_t cin, void* out, int dtype,
                            mla_stream_t stream);
/* f32 -> bf16 copy (nn.Linear weights for the bf16 GEMMs). */
int mla_convert_f32(const float* in, void* out, int64_t n, int dtype, mla_stream_t stream);
/* bf16 -> f32 copy (bf16 conv bottlenecks feeding the f32 MLA head, model.py:162-167 path). */
int mla_convert_bf16_to_f32(const void* in, float* out, int64_t n, mla_stream_t stream);
/* MLA_BF16X3 operand preparation: float32 (rows, cols) -> bf16 planes per segment of `seg` columns, [hi | lo] (copies = 2:
 * activations) or [hi | lo | hi] (copies = 3: weights; conv weights: rows = Cout*9 of the (Cout, 9, Cin) repack, seg = 64 = one K chunk;
 * Linear weights: seg = the activation's plane length, 512 after the conv stack's NHWC flatten, else in_features). */
int mla_split_bf16x3(const float* in, int64_t rows, int64_t cols, int64_t ld_in, void* out, int64_t ld_out, int64_t seg,
                     int copies, mla_stream_t stream);
/* split (rows, 2 cols) = [hi | lo] per segment -> float32 (rows, cols) = hi + lo (the bottleneck features of
 * just_bottlenecks=True, model.py:162-167, handed to the float32 head). */
int mla_merge_bf16x3(const void* in, int64_t rows, int64_t cols, int64_t ld_in, int64_t seg, float* out, mla_stream_t stream);
/* nn.Linear in the MLA_BF16X3 mode (vggish.py:13-19 at f32-grade accuracy on the bf16 matrix cores): a (M, 2K) split
 * activations, w (N, 3K) from mla_split_bf16x3, K = in_features, seg as above; out_dtype MLA_F32 (M, N) or MLA_BF16X3
 * (M, 2N) = [hi(N) | lo(N)]. */
int mla_linear_bf16x3(const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias, void* out, int64_t ldo,
                      int64_t M, int64_t N, int64_t K, int64_t seg, int out_dtype, int relu, mla_stream_t stream);

/* features[0..2]: Conv2d(1, 64, 3, pad 1) + ReLU + MaxPool2d(2, 2) fused.
 * x: (n, 96, 64) examples (x_dtype MLA_F32 | MLA_BF16); w: (64, 1, 3, 3) f32; bias (64) f32;
 * out: (n, 48, 32, 64) NHWC in `dtype`. */
int mla_vggish_conv1(const void* x, int x_dtype, int64_t n, const float* w, const float* bias, void* out,
                     int dtype, mla_stream_t stream);

/* conv `layer` in 2..6 = features[3], [6], [8], [11], [13], each fused with its ReLU and,
 * for layers 2, 4 and 6, with the MaxPool2d(2, 2) that follows it:
 *   2: (n,48,32, 64) -> (n,24,16,128)     3: (n,24,16,128) -> (n,24,16,256)
 *   4: (n,24,16,256) -> (n,12, 8,256)     5: (n,12, 8,256) -> (n,12, 8,512)
 *   6: (n,12, 8,512) -> (n, 6, 4,512)
 * in / out / w_repacked (from mla_conv_repack_weights) in `dtype`; bias f32. */
int mla_vggish_conv(int layer, const void* in, const void* w_repacked, const float* bias, void* out,
                    int64_t n, int dtype, mla_stream_t stream);

/* torch.nn.Linear (+ optional ReLU): out[M, N] = act(a[M, K] . w[N, K]^T + bias[N]).
 * VGG.embeddings (vggish.py:13-19) and the MLA fc / fcv layers (model.py:207-210, :230).
 * a, w in `dtype` with leading dimensions lda, ldw (elements, 16-byte multiples); out in
 * out_dtype (MLA_F32, or MLA_BF16 when dtype is MLA_BF16); bias f32 or NULL. */
int mla_linear(const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias, void* out,
               int64_t ldo, int64_t M, int64_t N, int64_t K, int dtype, int out_dtype, int relu,
               mla_stream_t stream);
/* mla_linear for NARROW forward layers with a long reduction (vggish.py:17: Linear(4096, 128)): K is cut into `splits` equal
 * ranges of whole 128-byte stages that run as separate workgroups; the float32 partial sums (workspace: splits * M * N floats)
 * are added in range order, then bias / ReLU. The caller fixes `splits` per LAYER, never per batch: every addition's order
 * depends on (K, splits) only, so a row's result does not depend on the batch it is computed in. f32 or bf16 operands. */
int mla_linear_ksplit(const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias, void* out, int64_t ldo,
                      int64_t M, int64_t N, int64_t K, int dtype, int out_dtype, int relu, int splits, float* workspace,
                      int64_t workspace_floats, mla_stream_t stream);
/* mla_linear in f32 with the reduction dimension split over `splits` workgroup ranges (weight
 * gradients: few output tiles, K = batch rows); partial sums go through workspace
 * (splits * M * N floats) and are combined in fixed order. */
int mla_linear_splitk(const float* a, int64_t lda, const float* w, int64_t ldw, const float* bias, float* out,
                      int64_t ldo, int64_t M, int64_t N, int64_t K, int relu, int splits, float* workspace,
                      int64_t workspace_floats, mla_stream_t stream);
/* nn.Linear with a NARROW output (N <= 16; model.py:230 fcv: 600 -> 10): one pass over `a`, 16 lanes per row, weights in LDS.
 * f32; K % 4 == 0, rows 16-byte aligned, K * N * 4 <= 64 KiB. Every row is computed the same way whatever M. */
int mla_linear_narrow(const float* a, int64_t lda, const float* w, int64_t ldw, const float* bias, float* out,
                      int64_t ldo, int64_t M, int64_t N, int64_t K, mla_stream_t stream);
/* Same contract in f32 without alignment requirements, for tiny layers (model.py:255 fc). */
int mla_linear_small(const float* a, int64_t lda, const float* w, int64_t ldw, const float* bias,
                     float* out, int64_t ldo, int64_t M, int64_t N, int64_t K, mla_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Multi-level attention head: model.py:200-269 (f32 throughout)
 * ---------------------------------------------------------------------------------- */

/* torch.nn.BatchNorm1d batch statistics (train mode). x: (rows, cols) f32, leading dim ldx.
 *   mode 0: channel = row % period  -- BatchNorm1d(T) on a (B, T, F) tensor flattened to
 *           (B*T, F) (model.py:205, :213, :232-233): statistics over (batch, feature) per slot;
 *   mode 1: channel = column        -- BatchNorm1d(K) on (B, K) (model.py:256).
 * Writes mean and BIASED variance (what normalisation uses); if running_mean/var are non-NULL
 * and momentum >= 0 they are updated in place with the UNBIASED variance, as torch does.
 * workspace: mla_bn_stats_workspace_bytes() bytes of device memory. Deterministic. */
int64_t mla_bn_stats_workspace_bytes(void);
int mla_bn_stats(const float* x, int64_t rows, int64_t cols, int64_t ldx, int mode, int period,
                 void* workspace, float* mean, float* var_biased, float* running_mean,
                 float* running_var, float momentum, mla_stream_t stream);

/* y = act((x - mean[c]) / sqrt(var[c] + eps) * gamma[c] + beta[c]), then optional dropout:
 * y = keep_mask ? y * drop_scale : 0 (keep_mask: rows*cols bytes or NULL). act: 0 none, 1 ReLU
 * (model.py:219), 2 sigmoid (model.py:268). Channel modes as mla_bn_stats. In eval mode the
 * caller passes the running statistics, in train mode the batch statistics. */
int mla_bn_apply(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t rows, int64_t cols, int mode,
                 int period, const float* mean, const float* var, const float* gamma, const float* beta,
                 float eps, int act, const uint8_t* keep_mask, float drop_scale, mla_stream_t stream);

/* model.AttentionModule.forward (model.py:236-242) after the fcv Linear: z (bags*T, K) ->
 * y (bags, K) written with leading dimension ldy (so levels concatenate in place,
 * model.py:267). BatchNorm parameters of normv (v_*) and normf (f_*) per time slot (T values).
 * att_out / cla_out (bags*T*K each) receive softmax / sigmoid for the backward pass, or NULL. */
int mla_attention_pool(const float* z, int64_t bags, int T, int K, const float* v_mean, const float* v_var,
                       const float* v_gamma, const float* v_beta, const float* f_mean, const float* f_var,
                       const float* f_gamma, const float* f_beta, float eps, float* y, int64_t ldy,
                       float* att_out, float* cla_out, mla_stream_t stream);

/* Two-stage form of mla_bn_stats for data-parallel training: stage 1 writes the LOCAL
 * per-channel (sum x, sum x^2) as 2*channels doubles; the host side may all-reduce them over
 * ranks (SyncBN: the reference's single process sees the global batch); stage 2 turns sums +
 * per-channel element count into mean / biased variance (+ running update). */
int mla_bn_stats_sums(const float* x, int64_t rows, int64_t cols, int64_t ldx, int mode, int period,
                      void* workspace, double* sums, mla_stream_t stream);
int mla_bn_stats_finish(const double* sums, int channels, double count, float* mean, float* var_biased,
                        float* running_mean, float* running_var, float momentum, int64_t* num_batches_tracked /* += 1, or NULL */,
                        mla_stream_t stream);
/* Both stages for a single process (no all-reduce in between) in two launches instead of three; same results bit for bit.
 * sums_out: the 2*channels doubles of stage 1 (for a second BatchNorm fed by the same tensor, model.py:237-238). */
int mla_bn_stats_fused(const float* x, int64_t rows, int64_t cols, int64_t ldx, int mode, int period, void* workspace,
                       double* sums_out, float* mean, float* var_biased, float* running_mean, float* running_var,
                       float momentum, int64_t* num_batches_tracked, mla_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Training step: train.py:124-138 (zero_grad, forward, CrossEntropyLoss, backward, Adam)
 * ---------------------------------------------------------------------------------- */

/* BatchNorm1d backward (train mode), through the activation/dropout fused in mla_bn_apply:
 * g = dy taken through act (1: ReLU[+dropout] using the forward output yout, 2: sigmoid, 0: none).
 * Stage 1: LOCAL per-channel (sum g, sum g*xhat) as 2*channels doubles (all-reducible).
 * Stage 2: dx = gamma*inv*(g - sum g/N - xhat*sum(g xhat)/N) from the GLOBAL sums and count N
 * (written, or added when accumulate != 0; NULL to skip) and dgamma/dbeta from the LOCAL sums
 * (the gradient all-reduce adds the ranks' parts; NULL to skip). */
int mla_bn_bwd_sums(const float* x, int64_t ldx, const float* dy, int64_t ld_dy, const float* yout, int64_t ld_y,
                    int act, float drop_scale, int64_t rows, int64_t cols, int mode, int period, const float* mean,
                    const float* var, float eps, void* workspace, double* sums, mla_stream_t stream);
int mla_bn_bwd_apply(const float* x, int64_t ldx, const float* dy, int64_t ld_dy, const float* yout, int64_t ld_y,
                     int act, float drop_scale, int64_t rows, int64_t cols, int mode, int period, const float* mean,
                     const float* var, const float* gamma, float eps, const double* sums_global,
                     const double* sums_local, double count, float* dx, int64_t ld_dx, int accumulate,
                     float* dgamma, float* dbeta, mla_stream_t stream);

/* Backward of mla_attention_pool: dy (bags, K; leading dim ld_dy) and the saved att / cla ->
 * gradients w.r.t. the normv output (du_v) and the normf output (du_f), (bags*T, K) each. */
int mla_attention_pool_bwd(const float* dy, int64_t ld_dy, const float* att, const float* cla, int64_t bags, int T,
                           int K, float* du_v, float* du_f, mla_stream_t stream);

/* Backward of mla_linear_small: da (M, K) = dz . w, dw (N, K) = dz^T . a, db (N) = column sums. */
int mla_linear_small_bwd(const float* a, int64_t lda, const float* w, int64_t ldw, const float* dz, int64_t ldz,
                         int64_t M, int64_t N, int64_t K, float* da, int64_t ldda, float* dw, float* db,
                         mla_stream_t stream);

/* out[c][r] = in[r][c] (f32). The Linear backward feeds the K-contiguous MFMA GEMM with
 * transposed copies: dX = mla_linear(dZ, W^T), dW = mla_linear(dZ^T, X^T). */
int mla_transpose_f32(const float* in, int64_t ld_in, float* out, int64_t ld_out, int64_t rows, int64_t cols,
                      mla_stream_t stream);
/* out[c] = sum_r x[r][c] (bias gradients). workspace: 64 * cols doubles. Deterministic. */
int mla_col_sum(const float* x, int64_t ldx, int64_t rows, int64_t cols, void* workspace, float* out,
                mla_stream_t stream);
/* y += a * x */
int mla_axpy(float a, const float* x, float* y, int64_t n, mla_stream_t stream);

/* nn.CrossEntropyLoss (train.py:372) on scores x (rows, K) with int64 labels: writes
 * loss = inv_total * sum_b (logsumexp(x_b) - x_b[y_b]) and, if dx != NULL,
 * dx = inv_total * (softmax(x_b) - onehot(y_b)); n_correct (optional, TWO ints): [0] = #argmax hits
 * (train.py:133), [1] = #labels outside [0, K) -- those rows contribute neither loss nor gradient, and the loss is NaN
 * when there is one (nn.CrossEntropyLoss raises; the caller does, on reading [1]). Two separate counters so that a sum
 * over data-parallel ranks cannot cancel one against the other. inv_total = 1 / global batch (mean over all ranks). */
int mla_cross_entropy(const float* x, int64_t ldx, const int64_t* labels, int64_t rows, int K, float inv_total,
                      float* loss, float* dx, int64_t ld_dx, int* n_correct, mla_stream_t stream);

/* --- finetune: gradients of the VGGish feature stack (train.py:96-97, :137), f32, NHWC ------ */

/* Generic 3x3/pad-1 convolution entry over the compiled shape set (VGGish forward with or
 * without the fused 2x2 pool, and the five dgrad shapes): act != 0 -> bias + ReLU epilogue,
 * act == 0 -> plain store (transposed convolution; bias may be NULL). dtype MLA_F32 or MLA_BF16
 * (activations and repacked weights in that type, f32 accumulation). */
int mla_conv3x3(const void* in, const void* w_packed, const float* bias, void* out, int64_t n, int H, int W,
                int cin, int cout, int pool, int act, int dtype, mla_stream_t stream);
/* Training forward of a POOLED layer (conv2 / conv4 / conv6 shapes): bias + ReLU, writes BOTH the pre-pool activation
 * (n, H, W, cout) -- kept for the pool / ReLU backward -- and its 2x2 max-pool (n, H/2, W/2, cout) from one pass (the separate
 * mla_maxpool2x2 over the tensor just written is not needed). dtype MLA_F32 or MLA_BF16. */
int mla_conv3x3_train(const void* in, const void* w_packed, const float* bias, void* out_prepool, void* out_pooled, int64_t n,
                      int H, int W, int cin, int cout, int dtype, mla_stream_t stream);
/* (Cout, Cin, 3, 3) -> (Cin, 9, Cout), taps flipped: weights of the dgrad convolution. */
int mla_conv_repack_dgrad(const float* w_oihw, int64_t cout, int64_t cin, float* out, mla_stream_t stream);
/* nn.MaxPool2d(2, 2) on a kept NHWC activation (n, H, W, C) -> (n, H/2, W/2, C). */
int mla_maxpool2x2(const float* a, float* out, int64_t n, int H, int W, int C, mla_stream_t stream);
/* dZ (n, H, W, C) from the gradient of the layer output: pool != 0: d_out is (n, H/2, W/2, C),
 * routed to the first maximum of each window of `a` (kept pre-pool, post-ReLU activation) and
 * masked by ReLU; pool == 0: dZ = d_out * (a > 0). */
int mla_relu_pool_bwd(const float* a, const float* d_out, float* dz, int64_t n, int H, int W, int C, int pool,
                      mla_stream_t stream);
/* the same plus the layer's bias gradient db[C] = column sums of dZ (nn.Conv2d bias, loss.backward() train.py:137), summed
 * on the way in double precision; workspace: mla_relu_pool_bwd_bias_workspace_bytes() bytes */
int64_t mla_relu_pool_bwd_bias_workspace_bytes(void);
int mla_relu_pool_bwd_bias(const float* a, const float* d_out, float* dz, int64_t n, int H, int W, int C, int pool,
                           void* workspace, float* db, mla_stream_t stream);
/* dW (Cout, Cin, 3, 3) = sum over pixels of dZ (n,H,W,Cout) x shifted a_in (n,H,W,Cin); f32 MFMA,
 * deterministic split reduction. workspace: mla_conv_wgrad_workspace_floats() floats. */
int64_t mla_conv_wgrad_workspace_floats(void);
int mla_conv_wgrad(const float* dz, const float* a_in, int64_t n, int H, int W, int cin, int cout, float* workspace,
                   int64_t workspace_floats, float* dw_oihw, mla_stream_t stream);
/* Backward of features[0..2] (Conv2d(1,64)+ReLU+MaxPool): x (n,96,64), d_pooled (n,48,32,64) ->
 * dw (64,1,3,3), db (64). workspace: 1024*8*80 floats. */
int mla_conv1_bwd(const float* x, const float* w, const float* bias, const float* d_pooled, int64_t n, float* workspace,
                  float* dw, float* db, mla_stream_t stream);

/* --- the same backward in bf16 (activations and gradients bf16, f32 accumulation, f32 weight / bias gradients):
 *     what the finetune step runs in at speed; weights stay f32 masters, bf16 copies are re-derived after each Adam step --- */
int mla_conv_repack_dgrad_bf16(const float* w_oihw, int64_t cout, int64_t cin, void* out_bf16, mla_stream_t stream);
int mla_maxpool2x2_bf16(const void* a, void* out, int64_t n, int H, int W, int C, mla_stream_t stream);
/* dZ (bf16) as mla_relu_pool_bwd; a / d_out both bf16 or both f32 (the last Linear's f32 output). db != NULL: also the bias
 * gradient (column sums of dZ, double-precision partials in `workspace` of mla_relu_pool_bwd_bf16_workspace_bytes()). */
int64_t mla_relu_pool_bwd_bf16_workspace_bytes(void);
int mla_relu_pool_bwd_bf16(const void* a, int a_dtype, const void* d_out, int d_dtype, void* dz, int64_t n, int H, int W,
                           int C, int pool, void* workspace, float* db, mla_stream_t stream);
/* dW f32 (Cout, Cin, 3, 3) from bf16 dZ and bf16 a_in on v_mfma_f32_16x16x32_bf16 (operands through the transposing LDS
 * read ds_read_b64_tr_b16); workspace as mla_conv_wgrad. */
int mla_conv_wgrad_bf16(const void* dz, const void* a_in, int64_t n, int H, int W, int cin, int cout, float* workspace,
                        int64_t workspace_floats, float* dw_oihw, mla_stream_t stream);
/* conv1 backward with the incoming gradient in bf16: recompute and weight-gradient products on the matrix cores (patch GEMM, see
 * cnn_train_bf16.hip). workspace: mla_conv1_bwd_workspace_floats() floats (also enough for mla_conv1_bwd). */
/* Training forward of a pooled layer, compact form: instead of the pre-pool activation of mla_conv3x3_train, one BYTE per pooled element
 * (N, H/2, W/2, Cout) -- the window position 0..3 of the first maximum in nn.MaxPool2d's order, or 4 where the ReLU is off -- which is all
 * autograd's backward of `MaxPool2d(ReLU(conv))` (vggish.py:108-118 under train.py:137) needs from that tensor; and the backward that
 * consumes it: dZ (N, H, W, C) bf16 from the codes and the pooled gradient, db on the way (workspace as mla_relu_pool_bwd_bf16). */
int mla_conv3x3_train_codes(const void* in, const void* w_packed, const float* bias, void* out_codes_u8, void* out_pooled, int64_t n,
                            int H, int W, int cin, int cout, int dtype, mla_stream_t stream);
int mla_pool_bwd_codes_bf16(const void* codes_u8, const void* d_pooled_bf16, void* dz_bf16, int64_t n, int H, int W, int C, void* workspace,
                            float* db, mla_stream_t stream);
int mla_conv1_bwd_bf16(const float* x, const float* w, const float* bias, const void* d_pooled_bf16, int64_t n, float* workspace,
                       float* dw, float* db, mla_stream_t stream);
int64_t mla_conv1_bwd_workspace_floats(void);
/* (rows, cols) bf16 -> (cols, ld_out >= rows) bf16, the padding columns zeroed: K-contiguous operands of the Linear backward */
int mla_transpose_bf16(const void* in, int64_t ld_in, void* out, int64_t ld_out, int64_t rows, int64_t cols, mla_stream_t stream);
/* column sums of a bf16 (rows, cols) matrix -> f32 (bias gradient of a Linear); workspace: 64 * cols doubles */
int mla_col_sum_bf16(const void* x, int64_t ldx, int64_t rows, int64_t cols, void* workspace, float* out, mla_stream_t stream);

/* torch.optim.Adam step t (train.py:369; no weight decay, no amsgrad) over one flat buffer. */
int mla_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                  float eps, int64_t step, mla_stream_t stream);
/* The same step with its two step-dependent scalars read from device memory (scal2_dev[0] = lr / (1 - beta1^t), [1] = 1 / sqrt(1 -
 * beta2^t)), so that a HIP graph holding the launch replays train.py:138 with the step count of the moment (the host form bakes
 * them into the launch). mla_adam_prepare writes them for step t, computed on the host exactly as mla_adam_step does: enqueue it
 * in front of every replay. */
int mla_adam_prepare(float* scal2_dev, float lr, float beta1, float beta2, int64_t step, mla_stream_t stream);
int mla_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, float beta1, float beta2, float eps,
                      const float* scal2_dev, mla_stream_t stream);
/* *counter_dev += delta: the call counter mla_dropout_mask_dev reads, advanced once per step from inside the graph */
int mla_counter_add(int64_t* counter_dev, int64_t delta, mla_stream_t stream);

/* vggish_input.py:52-53 `resampy.resample(data, sample_rate, 16000)`: band-limited sinc interpolation (resampy/interpn.py)
 * of a mono float32 waveform. win / delta: DEVICE tables of the interpolation filter in double precision (right half of the
 * windowed sinc, `num_table` entries per zero crossing, already scaled by the ratio when it is < 1; delta[i] = win[i+1] - win[i],
 * 0 for the last) -- host-side setup like the mel matrix (the package builds resampy's published 'kaiser_best' design).
 * n_out must equal mla_resample_length() = int(n_in * sr_out / sr_in); MLA_E_SHORT if that is 0 (resampy raises ValueError).
 * Parity with resampy itself is UNPINNED (the dependency is absent here): DESIGN.md section 5. */
int64_t mla_resample_length(int64_t n_in, double sr_in, double sr_out);
int mla_resample(const float* x, int64_t n_in, double sr_in, double sr_out, const double* win, const double* delta, int nwin,
                 int num_table, float* y, int64_t n_out, mla_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Data-parallel exchange (the reference is single-process, train.py:119-142: these entry
 * points are what makes its step run sharded over the GPUs of a node) and the dropout masks of the step.
 * ---------------------------------------------------------------------------------- */

/* RCCL communicator of this process' rank. The library binds the RCCL ALREADY LOADED in the process (dlopen NOLOAD: the one
 * PyTorch ships, or the host's own; ROCm's librccl.so.1 only if none is loaded) -- it never links a second one.
 * mla_comm_unique_id: rank 0 fills a 128-byte HOST buffer (ncclGetUniqueId) and distributes it by the host's own means;
 * mla_comm_init_rank: every rank, with its HIP device current (ncclCommInitRank; blocks until all ranks have called);
 * an ncclComm_t the host created itself may be passed as `comm` to mla_allreduce_flat unchanged. */
int         mla_comm_unique_id(void* id_host_128);
int         mla_comm_init_rank(void** comm_out, int nranks, const void* id_host_128, int rank);
int         mla_comm_destroy(void* comm);
int         mla_comm_count(void* comm, int* count);   /* ncclCommCount: the number of ranks the communicator spans */
const char* mla_comm_library_origin(void);   /* "already loaded by the host" | "loaded by libmla_hip" | "none" */

/* In-place sum over the ranks of `comm` of a flat device buffer (ncclAllReduce, ncclSum), enqueued on `stream`:
 * the gradient buffer (or one bucket of it) of loss.backward() (train.py:137) before optimizer.step() (train.py:138),
 * the SyncBN (sum, sum of squares) doubles of mla_bn_stats_sums / mla_bn_bwd_sums, the loss and n_correct.
 * dtype: MLA_F32, MLA_F64 or MLA_I32. */
int mla_allreduce_flat(void* buf, int64_t count, int dtype, void* comm, mla_stream_t stream);

/* nn.Dropout(p) keep-mask (model.py:213, 1 = keep), counter-based so that it is reproducible and shardable: element i
 * of the GLOBAL tensor gets 1 iff hash24(seed, stream_id, offset + i) >= round(p_drop * 2^24), hash24 = top 24 bits of
 * the splitmix64 finaliser over (seed, stream_id, index) (bit-exact numpy restatement: <pkg>/weights.py keep_mask).
 * A rank that owns elements [offset, offset + n) of the global batch passes its offset: N ranks draw the masks of a
 * single-process run. Consumed by mla_bn_apply (keep_mask argument). */
int mla_dropout_mask(uint8_t* mask, int64_t n, uint64_t seed, uint64_t stream_id, uint64_t offset, float p_drop,
                     mla_stream_t stream);
/* The same mask with stream_id = stream_base + *counter_dev + 1 read on the device: inside a HIP graph of the training step
 * every replay draws the next mask of the module's sequence (model.py:213 draws a fresh one per forward). */
int mla_dropout_mask_dev(uint8_t* mask, int64_t n, uint64_t seed, uint64_t stream_base, const int64_t* counter_dev,
                         uint64_t offset, float p_drop, mla_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MLA_HIP_H */
