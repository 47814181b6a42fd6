// mla_head.hip -- the small, HBM/launch-bound pieces of the multi-level-attention head
// (reference: model.py:200-269), forward and backward. The Linear layers go through gemm.hip.
//
//   bn_stats / bn_apply   torch.nn.BatchNorm1d as the reference uses it (model.py:205, :213,
//   bn_bwd_*              :232-233, :256). On a (B, T, F) tensor BatchNorm1d(T) takes the TIME
//                         SLOT as channel: statistics over (batch, feature) per t ("row-periodic"
//                         mode on the flattened (B*T, F) matrix, channel = row % T); on the (B, K)
//                         logits BatchNorm1d(K) is per column. Train mode: biased batch variance
//                         for normalisation, unbiased for the running update (momentum 0.1).
//                         bn_apply fuses the affine, ReLU / sigmoid and the dropout mask; the
//                         backward takes the gradient back through the same fused tail.
//                         Statistics are split in two stages (local sums -> finish) so that
//                         data-parallel ranks can all-reduce the sums in between (SyncBN).
//   attention_pool(_bwd)  model.py:237-240: att = softmax_K(BNv(z)), cla = sigmoid(BNf(z)) on the
//                         SAME z = fcv(h) (the reference never uses fcf), att normalised over T,
//                         y = sum_T cla * att. 16 lanes per bag, shuffle reductions.
//   linear_small(_bwd)    model.py:255/:268 fc (L*K -> K): too small / unaligned for the MFMA GEMM.
//
// Reductions are deterministic: per-block double-precision partials in a caller-provided
// workspace, combined in fixed order by a single finishing block.
#include "common.h"

namespace {

constexpr int kStatBlocks = 512;        // upper bound on partial-producing blocks
constexpr int kMaxChannels = 64;

__device__ __forceinline__ double wave_sum(double v) {
    _Pragma("unroll") for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ float sigmoidf(float v) { return 1.f / (1.f + __expf(-v)); }

// gradient entering the BatchNorm output, taken back through the fused activation / dropout
__device__ __forceinline__ float grad_through_act(float dy, float yout, int act, float drop_scale) {
    if (act == 1) return yout > 0.f ? dy * drop_scale : 0.f;    // ReLU (+ dropout): yout = keep ? relu * scale : 0
    if (act == 2) return dy * yout * (1.f - yout);              // sigmoid
    return dy;
}

// element functors of the per-channel reductions: (row, col, channel) -> the two summands
typedef float f32x4_t __attribute__((ext_vector_type(4)));

struct StatsOp {                        // forward statistics: (x, x^2)
    const float* x; int64_t ldx;
    __device__ __forceinline__ void operator()(int64_t r, int c, int, double& a, double& b) const {
        const double v = x[r * ldx + c];
        a = v; b = v * v;
    }
    // four consecutive columns with 16-byte loads, added to (a, b) in column order
    __device__ __forceinline__ void quad(int64_t r, int c, int, double& a, double& b) const {
        const f32x4_t v = *reinterpret_cast<const f32x4_t*>(x + r * ldx + c);
        _Pragma("unroll") for (int e = 0; e < 4; ++e) { const double d = v[e]; a += d; b += d * d; }
    }
    __device__ __forceinline__ bool vec_ok(int cols) const { return cols % 4 == 0 && ldx % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0; }
};

struct BwdOp {                          // backward: (g, g * xhat)
    const float *x, *dy, *yout, *mean, *var;
    int64_t ldx, ld_dy, ld_y;
    int act;
    float drop_scale, eps;
    __device__ __forceinline__ void operator()(int64_t r, int c, int ch, double& a, double& b) const {
        const float g = grad_through_act(dy[r * ld_dy + c], yout ? yout[r * ld_y + c] : 0.f, act, drop_scale);
        const float xhat = (x[r * ldx + c] - mean[ch]) * (1.0f / sqrtf(var[ch] + eps));
        a = g; b = double(g) * xhat;
    }
    __device__ __forceinline__ void quad(int64_t r, int c, int ch, double& a, double& b) const {
        const f32x4_t d = *reinterpret_cast<const f32x4_t*>(dy + r * ld_dy + c);
        const f32x4_t xv = *reinterpret_cast<const f32x4_t*>(x + r * ldx + c);
        f32x4_t y = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (yout) y = *reinterpret_cast<const f32x4_t*>(yout + r * ld_y + c);
        const float m = mean[ch], inv = 1.0f / sqrtf(var[ch] + eps);
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {
            const float g = grad_through_act(d[e], y[e], act, drop_scale);
            const float xhat = (xv[e] - m) * inv;
            a += g; b += double(g) * xhat;
        }
    }
    __device__ __forceinline__ bool vec_ok(int cols) const {
        return cols % 4 == 0 && ldx % 4 == 0 && ld_dy % 4 == 0 && (!yout || ld_y % 4 == 0) && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
               (reinterpret_cast<uintptr_t>(dy) & 15) == 0 && (reinterpret_cast<uintptr_t>(yout) & 15) == 0;
    }
};

// mode 0: channel = row % period; one wave per row, lanes stride the columns
template <typename Op>
__global__ __launch_bounds__(256) void sums_rows_kernel(Op op, int64_t rows, int cols, int period, double* __restrict__ partial) {
    __shared__ double part[4][kMaxChannels][2];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool vec = op.vec_ok(cols);
    for (int i = threadIdx.x; i < 4 * kMaxChannels * 2; i += 256) (&part[0][0][0])[i] = 0.0;
    __syncthreads();
    const int64_t per_block = ((rows + gridDim.x - 1) / gridDim.x + period - 1) / period * period;
    const int64_t r0 = int64_t(blockIdx.x) * per_block;
    const int64_t r1 = r0 + per_block < rows ? r0 + per_block : rows;
    for (int64_t r = r0 + wave; r < r1; r += 4) {
        const int chan = int(r % period);
        double s = 0.0, ss = 0.0;
        if (vec) {                                           // 16-byte loads: a wave-instruction covers 1 KiB of the row
            for (int c = 4 * lane; c < cols; c += 256) op.quad(r, c, chan, s, ss);
        } else {
            for (int c = lane; c < cols; c += 64) {
                double a, b;
                op(r, c, chan, a, b);
                s += a;
                ss += b;
            }
        }
        s = wave_sum(s);
        ss = wave_sum(ss);
        if (lane == 0) {
            part[wave][chan][0] += s;
            part[wave][chan][1] += ss;
        }
    }
    __syncthreads();
    if (threadIdx.x < period) {
        double s = 0.0, ss = 0.0;
        for (int w = 0; w < 4; ++w) { s += part[w][threadIdx.x][0]; ss += part[w][threadIdx.x][1]; }
        partial[(int64_t(blockIdx.x) * period + threadIdx.x) * 2] = s;
        partial[(int64_t(blockIdx.x) * period + threadIdx.x) * 2 + 1] = ss;
    }
}

// mode 1: channel = column (cols <= 64); thread (g, c) walks rows g, g + G, ...
template <typename Op>
__global__ __launch_bounds__(256) void sums_cols_kernel(Op op, int64_t rows, int cols, double* __restrict__ partial) {
    __shared__ double part[256][2];
    const int groups = 256 / cols, g = threadIdx.x / cols, c = threadIdx.x % cols;
    double s = 0.0, ss = 0.0;
    if (g < groups) {
        const int64_t per_block = (rows + gridDim.x - 1) / gridDim.x;
        const int64_t r0 = int64_t(blockIdx.x) * per_block;
        const int64_t r1 = r0 + per_block < rows ? r0 + per_block : rows;
        for (int64_t r = r0 + g; r < r1; r += groups) {
            double a, b;
            op(r, c, c, a, b);
            s += a;
            ss += b;
        }
    }
    part[threadIdx.x][0] = s;
    part[threadIdx.x][1] = ss;
    __syncthreads();
    if (threadIdx.x < cols) {
        double a = 0.0, b = 0.0;
        for (int gg = 0; gg < groups; ++gg) { a += part[gg * cols + threadIdx.x][0]; b += part[gg * cols + threadIdx.x][1]; }
        partial[(int64_t(blockIdx.x) * cols + threadIdx.x) * 2] = a;
        partial[(int64_t(blockIdx.x) * cols + threadIdx.x) * 2 + 1] = b;
    }
}

// fixed-order combination of the per-block partials -> sums[channel][2]: one wave per channel, lane l
// adds blocks l, l + 64, ... then a shuffle tree (a single thread walking 512 strided partials per
// channel took 110 us per call and 16 % of the training step)
__global__ __launch_bounds__(64) void partial_reduce_kernel(const double* __restrict__ partial, int blocks, int channels,
                                                            double* __restrict__ sums) {
    const int c = blockIdx.x, lane = threadIdx.x;
    double s = 0.0, ss = 0.0;
    for (int b = lane; b < blocks; b += 64) { s += partial[(int64_t(b) * channels + c) * 2]; ss += partial[(int64_t(b) * channels + c) * 2 + 1]; }
    s = wave_sum(s);
    ss = wave_sum(ss);
    if (lane == 0) { sums[2 * c] = s; sums[2 * c + 1] = ss; }
}

// partial_reduce_kernel and stats_finish_kernel in ONE launch for up to 16 channels (every BatchNorm of the head has 10): wave c
// combines channel c's partials in the same fixed order, its lane 0 writes sums[c] and finishes mean / biased variance / running
// statistics exactly as stats_finish_kernel does (same expressions: same bits). Used when no rank has to all-reduce the sums in between.
struct FinishArgs {
    double count; float* mean; float* var; float* run_mean; float* run_var; float momentum; int64_t* tracked;
};
__global__ __launch_bounds__(1024) void reduce_finish_kernel(const double* __restrict__ partial, int blocks, int channels,
                                                             double* __restrict__ sums, FinishArgs f) {
    const int c = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x == 0 && f.tracked) *f.tracked += 1;
    if (c >= channels) return;
    double s = 0.0, ss = 0.0;
    for (int b = lane; b < blocks; b += 64) { s += partial[(int64_t(b) * channels + c) * 2]; ss += partial[(int64_t(b) * channels + c) * 2 + 1]; }
    s = wave_sum(s);
    ss = wave_sum(ss);
    if (lane != 0) return;
    sums[2 * c] = s;
    sums[2 * c + 1] = ss;
    const double m = s / f.count;
    double v = ss / f.count - m * m;
    if (v < 0.0) v = 0.0;
    f.mean[c] = float(m);
    f.var[c] = float(v);
    if (f.run_mean && f.momentum >= 0.f) {
        const double unbiased = f.count > 1.0 ? v * f.count / (f.count - 1.0) : v;
        f.run_mean[c] = float((1.0 - f.momentum) * f.run_mean[c] + f.momentum * m);
        f.run_var[c] = float((1.0 - f.momentum) * f.run_var[c] + f.momentum * unbiased);
    }
}

template <typename Op>
int channel_sums(Op op, int64_t rows, int64_t cols, int mode, int period, void* workspace, double* sums, hipStream_t s,
                 const FinishArgs* finish = nullptr) {
    double* partial = static_cast<double*>(workspace);
    const int channels = mode == 0 ? period : int(cols);
    int blocks;
    if (mode == 0) {
        const int64_t groups = rows / period;
        blocks = int(groups < kStatBlocks ? groups : kStatBlocks);
        hipLaunchKernelGGL(sums_rows_kernel<Op>, dim3(blocks), dim3(256), 0, s, op, rows, int(cols), period, partial);
    } else {
        blocks = int((rows + 63) / 64 < kStatBlocks ? (rows + 63) / 64 : kStatBlocks);
        hipLaunchKernelGGL(sums_cols_kernel<Op>, dim3(blocks), dim3(256), 0, s, op, rows, int(cols), partial);
    }
    MLA_LAUNCH_OK("channel sums");
    if (finish && channels <= 16) {
        hipLaunchKernelGGL(reduce_finish_kernel, dim3(1), dim3(64 * channels), 0, s, partial, blocks, channels, sums, *finish);
        MLA_LAUNCH_OK("channel sums reduce + finish");
        return 1;                                           // finished as well
    }
    hipLaunchKernelGGL(partial_reduce_kernel, dim3(channels), dim3(64), 0, s, partial, blocks, channels, sums);
    MLA_LAUNCH_OK("channel sums reduce");
    return MLA_OK;
}

__global__ void stats_finish_kernel(const double* __restrict__ sums, int channels, double count, float* __restrict__ mean,
                                    float* __restrict__ var, float* __restrict__ run_mean, float* __restrict__ run_var,
                                    float momentum, int64_t* __restrict__ num_batches_tracked) {
    const int c = threadIdx.x;
    if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;      // nn.BatchNorm1d's counter, advanced by the same launch
    if (c >= channels) return;
    const double m = sums[2 * c] / count;
    double v = sums[2 * c + 1] / count - m * m;
    if (v < 0.0) v = 0.0;
    mean[c] = float(m);
    var[c] = float(v);
    if (run_mean && momentum >= 0.f) {
        const double unbiased = count > 1.0 ? v * count / (count - 1.0) : v;
        run_mean[c] = float((1.0 - momentum) * run_mean[c] + momentum * m);
        run_var[c] = float((1.0 - momentum) * run_var[c] + momentum * unbiased);
    }
}

// y = act((x - mean[c]) * rsqrt(var[c] + eps) * gamma[c] + beta[c]) * keep * drop_scale
template <int MODE>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ y,
                                                       int64_t ldy, int64_t rows, int cols, int period,
                                                       const float* __restrict__ mean, const float* __restrict__ var,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float eps, int act, const uint8_t* __restrict__ keep,
                                                       float drop_scale) {
    const int64_t total = rows * cols;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < total; i += int64_t(gridDim.x) * 256) {
        const int64_t r = i / cols;
        const int c = int(i - r * cols);
        const int ch = MODE == 0 ? int(r % period) : c;
        const float inv = 1.0f / sqrtf(var[ch] + eps);
        float v = (x[r * ldx + c] - mean[ch]) * inv * gamma[ch] + beta[ch];
        if (act == 1) v = fmaxf(v, 0.f);
        else if (act == 2) v = sigmoidf(v);
        if (keep) v = keep[i] ? v * drop_scale : 0.f;
        y[r * ldy + c] = v;
    }
}

// dx = gamma * inv * (g - sum(g)/N - xhat * sum(g xhat)/N), written or accumulated.
// sums_global drive dx (all-reduced over ranks under SyncBN); dgamma/dbeta come from sums_local.
template <int MODE>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(BwdOp op, const float* __restrict__ gamma, int64_t rows, int cols,
                                                           int period, const double* __restrict__ sums_global,
                                                           const double* __restrict__ sums_local, double count,
                                                           float* __restrict__ dx, int64_t ld_dx, int accumulate,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int channels = MODE == 0 ? period : cols;
    if (blockIdx.x == 0 && threadIdx.x < channels && dgamma) {
        dbeta[threadIdx.x] = float(sums_local[2 * threadIdx.x]);
        dgamma[threadIdx.x] = float(sums_local[2 * threadIdx.x + 1]);
    }
    if (!dx) return;
    const int64_t total = rows * cols;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < total; i += int64_t(gridDim.x) * 256) {
        const int64_t r = i / cols;
        const int c = int(i - r * cols);
        const int ch = MODE == 0 ? int(r % period) : c;
        double g, gx;
        op(r, c, ch, g, gx);
        const float inv = 1.0f / sqrtf(op.var[ch] + op.eps);
        const float xhat = (op.x[r * op.ldx + c] - op.mean[ch]) * inv;
        const float sg = float(sums_global[2 * ch] / count), sgx = float(sums_global[2 * ch + 1] / count);
        const float v = gamma[ch] * inv * (float(g) - sg - xhat * sgx);
        float* o = dx + r * ld_dx + c;
        *o = accumulate ? *o + v : v;
    }
}

struct BnParams { const float *mean, *var, *gamma, *beta; };

// 16 lanes per bag; lane t < T owns time slot t, all K classes in registers (K <= 16)
template <int KMAX>
__global__ __launch_bounds__(256) void attention_pool_kernel(const float* __restrict__ z, int64_t bags, int T, int K,
                                                             BnParams nv, BnParams nf, float eps, float* __restrict__ y,
                                                             int64_t ldy, float* __restrict__ att_out,
                                                             float* __restrict__ cla_out) {
    const int t = threadIdx.x & 15;
    const int64_t bag = int64_t(blockIdx.x) * 16 + (threadIdx.x >> 4);
    const bool live = bag < bags && t < T;
    float att[KMAX], cla[KMAX];
    if (live) {
        const float* row = z + (bag * T + t) * K;
        const float iv = 1.0f / sqrtf(nv.var[t] + eps), jf = 1.0f / sqrtf(nf.var[t] + eps);
        float mx = -3.0e38f;
        _Pragma("unroll") for (int k = 0; k < KMAX; ++k)
            if (k < K) {
                const float v = row[k];
                att[k] = (v - nv.mean[t]) * iv * nv.gamma[t] + nv.beta[t];
                cla[k] = sigmoidf((v - nf.mean[t]) * jf * nf.gamma[t] + nf.beta[t]);
                mx = fmaxf(mx, att[k]);
            }
        float den = 0.f;
        _Pragma("unroll") for (int k = 0; k < KMAX; ++k)
            if (k < K) { att[k] = __expf(att[k] - mx); den += att[k]; }
        const float inv = 1.0f / den;
        _Pragma("unroll") for (int k = 0; k < KMAX; ++k)
            if (k < K) {
                att[k] *= inv;
                if (att_out) att_out[(bag * T + t) * K + k] = att[k];
                if (cla_out) cla_out[(bag * T + t) * K + k] = cla[k];
            }
    } else {
        _Pragma("unroll") for (int k = 0; k < KMAX; ++k) { att[k] = 0.f; cla[k] = 0.f; }
    }
    _Pragma("unroll") for (int k = 0; k < KMAX; ++k) {
        if (k < K) {                                   // K is block-uniform: shuffles stay convergent
            float s = att[k];
            _Pragma("unroll") for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 16);
            float num = live ? cla[k] * (att[k] / s) : 0.f;     // model.py:239-240: cla * (att / sum_T att)
            _Pragma("unroll") for (int o = 8; o > 0; o >>= 1) num += __shfl_xor(num, o, 16);
            if (t == 0 && bag < bags) y[bag * ldy + k] = num;
        }
    }
}

// backward of the pooling given dy (bags, K): gradients w.r.t. the two BatchNorm outputs
//   y = sum_t cla n,  n = att / S,  S = sum_t att
//   d cla = dy n ;  d att = dy (cla - y) / S
//   du_f = d cla * cla (1 - cla) ;  du_v = att * (d att - sum_k d att * att)      (softmax over k)
template <int KMAX>
__global__ __launch_bounds__(256) void attention_pool_bwd_kernel(const float* __restrict__ dy, int64_t ld_dy,
                                                                 const float* __restrict__ att_in, const float* __restrict__ cla_in,
                                                                 int64_t bags, int T, int K, float* __restrict__ du_v,
                                                                 float* __restrict__ du_f) {
    const int t = threadIdx.x & 15;
    const int64_t bag = int64_t(blockIdx.x) * 16 + (threadIdx.x >> 4);
    const bool live = bag < bags && t < T;
    float att[KMAX], cla[KMAX], datt[KMAX];
    _Pragma("unroll") for (int k = 0; k < KMAX; ++k) {
        att[k] = (live && k < K) ? att_in[(bag * T + t) * K + k] : 0.f;
        cla[k] = (live && k < K) ? cla_in[(bag * T + t) * K + k] : 0.f;
    }
    float dot = 0.f;
    _Pragma("unroll") for (int k = 0; k < KMAX; ++k) {
        datt[k] = 0.f;
        if (k < K) {
            float s = att[k];
            _Pragma("unroll") for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 16);
            float yk = live ? cla[k] * (att[k] / s) : 0.f;
            _Pragma("unroll") for (int o = 8; o > 0; o >>= 1) yk += __shfl_xor(yk, o, 16);
            if (live) {
                const float g = dy[bag * ld_dy + k];
                const float dcla = g * (att[k] / s);
                datt[k] = g * (cla[k] - yk) / s;
                du_f[(bag * T + t) * K + k] = dcla * cla[k] * (1.f - cla[k]);
                dot += datt[k] * att[k];
            }
        }
    }
    if (live) {
        _Pragma("unroll") for (int k = 0; k < KMAX; ++k)
            if (k < K) du_v[(bag * T + t) * K + k] = att[k] * (datt[k] - dot);
    }
}

__global__ __launch_bounds__(256) void linear_small_kernel(const float* __restrict__ a, int64_t lda,
                                                           const float* __restrict__ w, int64_t ldw,
                                                           const float* __restrict__ bias, float* __restrict__ out,
                                                           int64_t ldo, int64_t M, int N, int K) {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= M * N) return;
    const int64_t m = i / N;
    const int n = int(i - m * N);
    float acc = bias ? bias[n] : 0.f;
    for (int k = 0; k < K; ++k) acc = fmaf(a[m * lda + k], w[int64_t(n) * ldw + k], acc);
    out[m * ldo + n] = acc;
}

// The same for a SHORT reduction and a wide output (the input gradient of the attention modules' fcv: dz (rows, 10) . W (10, 600)):
// the weights are staged once per block in LDS as [k][n], a thread produces four consecutive outputs of one row (16-byte store);
// the K-long sum of every output runs in the same order as in linear_small_kernel (k ascending, one fma chain): same bits.
__global__ __launch_bounds__(256) void linear_small_wide_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ w,
                                                                int64_t ldw, const float* __restrict__ bias, float* __restrict__ out,
                                                                int64_t ldo, int64_t M, int N, int K) {
    extern __shared__ float s_w[];                        // [K][N]
    for (int i = threadIdx.x; i < N * K; i += 256) {
        const int n = i / K, kk = i - n * K;
        s_w[kk * N + n] = w[int64_t(n) * ldw + kk];
    }
    __syncthreads();
    const int quads = N / 4;
    const int64_t total = M * quads;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < total; i += int64_t(gridDim.x) * 256) {
        const int64_t m = i / quads;
        const int n = int(i - m * quads) * 4;
        float acc[4];
        _Pragma("unroll") for (int e = 0; e < 4; ++e) acc[e] = bias ? bias[n + e] : 0.f;
        for (int kk = 0; kk < K; ++kk) {
            const float av = a[m * lda + kk];
            const f32x4_t wv = *reinterpret_cast<const f32x4_t*>(s_w + kk * N + n);
            _Pragma("unroll") for (int e = 0; e < 4; ++e) acc[e] = fmaf(av, wv[e], acc[e]);
        }
        *reinterpret_cast<f32x4_t*>(out + m * ldo + n) = f32x4_t{acc[0], acc[1], acc[2], acc[3]};
    }
}

// out[m][n] = bias[n] + sum_k a[m][k] w[n][k] for a NARROW layer (N <= 16: the attention modules' fcv, model.py:230, 600 -> 10):
// HBM-bound on reading `a` once. 16 lanes per row; lane l takes the float4 pieces l, l + 16, ... of the row (a wave-instruction
// reads 4 x 256 contiguous bytes), the weights sit in LDS as [k/4][n] float4 so that the 16 lanes of a group read 16 different
// consecutive pieces; the N partial sums are reduced over the group's 16 lanes with four DPP/shuffle steps in a fixed order.
// An MFMA tile for a 10-column output leaves 80 of 256 CUs with work and is latency-bound on its serial K loop (49 us for
// 10 240 rows, the same for 1 020); this form is 6-8 us. Every row is computed the same way whatever the batch: bag-independent bits.
typedef float f4 __attribute__((ext_vector_type(4)));

template <int N>
__global__ __launch_bounds__(256) void linear_narrow_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ w,
                                                            int64_t ldw, const float* __restrict__ bias, float* __restrict__ out,
                                                            int64_t ldo, int64_t M, int K) {
    extern __shared__ __attribute__((aligned(16))) float sw[];           // [K/4][N] float4
    const int k4 = K / 4;
    for (int i = threadIdx.x; i < k4 * N; i += 256) {
        const int n = i % N, kk = i / N;
        reinterpret_cast<f4*>(sw)[kk * N + n] = *reinterpret_cast<const f4*>(w + int64_t(n) * ldw + 4 * kk);
    }
    __syncthreads();
    const int l = threadIdx.x & 15, grp = threadIdx.x >> 4;
    for (int64_t m = int64_t(blockIdx.x) * 16 + grp; m < M; m += int64_t(gridDim.x) * 16) {
        float acc[N];
        _Pragma("unroll") for (int n = 0; n < N; ++n) acc[n] = 0.f;
        const f4* row = reinterpret_cast<const f4*>(a + m * lda);
        for (int kk = l; kk < k4; kk += 16) {
            const f4 x = row[kk];
            _Pragma("unroll") for (int n = 0; n < N; ++n) {
                const f4 ww = reinterpret_cast<const f4*>(sw)[kk * N + n];
                acc[n] = fmaf(x.x, ww.x, acc[n]);
                acc[n] = fmaf(x.y, ww.y, acc[n]);
                acc[n] = fmaf(x.z, ww.z, acc[n]);
                acc[n] = fmaf(x.w, ww.w, acc[n]);
            }
        }
        _Pragma("unroll") for (int n = 0; n < N; ++n) {
            float v = acc[n];
            v += __shfl_xor(v, 8, 16);
            v += __shfl_xor(v, 4, 16);
            v += __shfl_xor(v, 2, 16);
            v += __shfl_xor(v, 1, 16);
            acc[n] = v;
        }
        if (l < N) {
            float v = 0.f;
            _Pragma("unroll") for (int n = 0; n < N; ++n) v = l == n ? acc[n] : v;
            out[m * ldo + l] = v + (bias ? bias[l] : 0.f);
        }
    }
}

// backward of out = a w^T + b for tiny N, K: da (M, K) = dz w ; dw (N, K) = dz^T a ; db (N) = sum_m dz
__global__ __launch_bounds__(256) void linear_small_bwd_kernel(const float* __restrict__ a, int64_t lda,
                                                               const float* __restrict__ w, int64_t ldw,
                                                               const float* __restrict__ dz, int64_t ldz, int64_t M, int N, int K,
                                                               float* __restrict__ da, int64_t ldda, float* __restrict__ dw,
                                                               float* __restrict__ db) {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < M * K) {
        const int64_t m = i / K;
        const int k = int(i - m * K);
        float acc = 0.f;
        for (int n = 0; n < N; ++n) acc = fmaf(dz[m * ldz + n], w[int64_t(n) * ldw + k], acc);
        da[m * ldda + k] = acc;
    }
    // dw and db: one WAVE per weight / bias element -- lanes stride over the rows, then a fixed-order butterfly over the lanes
    // (deterministic; one thread per weight walked all M rows alone: 122 us at 512 bags)
    const int64_t wv = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (wv < int64_t(N) * K + N) {
        double acc = 0.0;
        if (wv < int64_t(N) * K) {
            const int n = int(wv / K), k = int(wv % K);
            for (int64_t m = lane; m < M; m += 64) acc += double(dz[m * ldz + n]) * a[m * lda + k];
        } else {
            const int n = int(wv - int64_t(N) * K);
            for (int64_t m = lane; m < M; m += 64) acc += dz[m * ldz + n];
        }
        _Pragma("unroll") for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
        if (lane == 0) {
            if (wv < int64_t(N) * K) dw[wv] = float(acc);
            else db[wv - int64_t(N) * K] = float(acc);
        }
    }
}

}  // namespace

extern "C" int64_t mla_bn_stats_workspace_bytes(void) { return (int64_t(kStatBlocks) + 1) * kMaxChannels * 2 * sizeof(double); }

extern "C" int mla_bn_stats_sums(const float* x, int64_t rows, int64_t cols, int64_t ldx, int mode, int period, void* workspace,
                                 double* sums, mla_stream_t stream) {
    MLA_REQUIRE(x && workspace && sums && rows > 0 && cols > 0 && ldx >= cols, MLA_E_ARG, "bad bn_stats arguments");
    MLA_REQUIRE(mode == 0 || mode == 1, MLA_E_ARG, "bn_stats mode %d", mode);
    const int channels = mode == 0 ? period : int(cols);
    MLA_REQUIRE(channels >= 1 && channels <= kMaxChannels, MLA_E_SHAPE, "bn_stats supports 1..%d channels (got %d)", kMaxChannels, channels);
    MLA_REQUIRE(mode == 1 || rows % period == 0, MLA_E_SHAPE, "rows %lld not a multiple of period %d", (long long)rows, period);
    return channel_sums(StatsOp{x, ldx}, rows, cols, mode, period, workspace, sums, static_cast<hipStream_t>(stream));
}

extern "C" int mla_bn_stats_finish(const double* sums, int channels, double count, float* mean, float* var_biased,
                                   float* running_mean, float* running_var, float momentum, int64_t* num_batches_tracked,
                                   mla_stream_t stream) {
    MLA_REQUIRE(sums && mean && var_biased && channels >= 1 && channels <= kMaxChannels && count > 0, MLA_E_ARG, "bad bn_stats_finish arguments");
    hipLaunchKernelGGL(stats_finish_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), sums, channels, count, mean,
                       var_biased, running_mean, running_var, momentum, num_batches_tracked);
    MLA_LAUNCH_OK("bn stats finish");
    return MLA_OK;
}

extern "C" int mla_bn_stats(const float* x, int64_t rows, int64_t cols, int64_t ldx, int mode, int period, void* workspace,
                            float* mean, float* var_biased, float* running_mean, float* running_var, float momentum,
                            mla_stream_t stream) {
    MLA_REQUIRE(workspace, MLA_E_ARG, "null workspace");
    double* sums = static_cast<double*>(workspace) + int64_t(kStatBlocks) * kMaxChannels * 2;   // tail of the workspace
    const int rc = mla_bn_stats_sums(x, rows, cols, ldx, mode, period, workspace, sums, stream);
    if (rc != MLA_OK) return rc;
    const int channels = mode == 0 ? period : int(cols);
    const double count = mode == 0 ? double(rows / period) * double(cols) : double(rows);
    return mla_bn_stats_finish(sums, channels, count, mean, var_biased, running_mean, running_var, momentum, nullptr, stream);
}

// mla_bn_stats_sums + mla_bn_stats_finish in two launches instead of three (the reduction of the per-block partials and the finish
// share one kernel for <= 16 channels): the single-process form of the train-mode statistics. sums_out: 2 * channels doubles, as
// mla_bn_stats_sums writes them (a second BatchNorm fed by the same tensor finishes from them).
extern "C" int mla_bn_stats_fused(const float* x, int64_t rows, int64_t cols, int64_t ldx, int mode, int period, void* workspace,
                                  double* sums_out, float* mean, float* var_biased, float* running_mean, float* running_var,
                                  float momentum, int64_t* num_batches_tracked, mla_stream_t stream) {
    MLA_REQUIRE(x && workspace && sums_out && mean && var_biased && rows > 0 && cols > 0 && ldx >= cols, MLA_E_ARG, "bad bn_stats arguments");
    MLA_REQUIRE(mode == 0 || mode == 1, MLA_E_ARG, "bn_stats mode %d", mode);
    const int channels = mode == 0 ? period : int(cols);
    MLA_REQUIRE(channels >= 1 && channels <= kMaxChannels, MLA_E_SHAPE, "bn_stats supports 1..%d channels (got %d)", kMaxChannels, channels);
    MLA_REQUIRE(mode == 1 || rows % period == 0, MLA_E_SHAPE, "rows %lld not a multiple of period %d", (long long)rows, period);
    const double count = mode == 0 ? double(rows / period) * double(cols) : double(rows);
    const FinishArgs f{count, mean, var_biased, running_mean, running_var, momentum, num_batches_tracked};
    const int rc = channel_sums(StatsOp{x, ldx}, rows, cols, mode, period, workspace, sums_out, static_cast<hipStream_t>(stream), &f);
    if (rc == 1) return MLA_OK;
    if (rc != MLA_OK) return rc;
    return mla_bn_stats_finish(sums_out, channels, count, mean, var_biased, running_mean, running_var, momentum, num_batches_tracked, stream);
}

extern "C" int mla_bn_apply(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t rows, int64_t cols, int mode,
                            int period, const float* mean, const float* var, const float* gamma, const float* beta,
                            float eps, int act, const uint8_t* keep_mask, float drop_scale, mla_stream_t stream) {
    MLA_REQUIRE(x && y && mean && var && gamma && beta && rows >= 0 && cols > 0, MLA_E_ARG, "bad bn_apply arguments");
    MLA_REQUIRE((mode == 0 && period >= 1) || mode == 1, MLA_E_ARG, "bn_apply mode %d period %d", mode, period);
    MLA_REQUIRE(act >= 0 && act <= 2, MLA_E_ARG, "bn_apply act %d", act);
    if (rows == 0) return MLA_OK;
    const int64_t total = rows * cols;
    const unsigned grid = unsigned((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (mode == 0)
        hipLaunchKernelGGL(bn_apply_kernel<0>, dim3(grid), dim3(256), 0, s, x, ldx, y, ldy, rows, int(cols), period, mean, var,
                           gamma, beta, eps, act, keep_mask, drop_scale);
    else
        hipLaunchKernelGGL(bn_apply_kernel<1>, dim3(grid), dim3(256), 0, s, x, ldx, y, ldy, rows, int(cols), period, mean, var,
                           gamma, beta, eps, act, keep_mask, drop_scale);
    MLA_LAUNCH_OK("bn_apply");
    return MLA_OK;
}

extern "C" int mla_bn_bwd_sums(const float* x, int64_t ldx, const float* dy, int64_t ld_dy, const float* yout, int64_t ld_y,
                               int act, float drop_scale, int64_t rows, int64_t cols, int mode, int period, const float* mean,
                               const float* var, float eps, void* workspace, double* sums, mla_stream_t stream) {
    MLA_REQUIRE(x && dy && mean && var && workspace && sums && rows > 0 && cols > 0, MLA_E_ARG, "bad bn_bwd_sums arguments");
    MLA_REQUIRE(act == 0 || yout, MLA_E_ARG, "bn_bwd needs the forward output for act %d", act);
    MLA_REQUIRE((mode == 0 && period >= 1 && period <= kMaxChannels && rows % period == 0) || (mode == 1 && cols <= kMaxChannels),
                MLA_E_SHAPE, "bn_bwd channel layout");
    BwdOp op{x, dy, yout, mean, var, ldx, ld_dy, ld_y, act, drop_scale, eps};
    return channel_sums(op, rows, cols, mode, period, workspace, sums, static_cast<hipStream_t>(stream));
}

extern "C" int mla_bn_bwd_apply(const float* x, int64_t ldx, const float* dy, int64_t ld_dy, const float* yout, int64_t ld_y,
                                int act, float drop_scale, int64_t rows, int64_t cols, int mode, int period, const float* mean,
                                const float* var, const float* gamma, float eps, const double* sums_global,
                                const double* sums_local, double count, float* dx, int64_t ld_dx, int accumulate,
                                float* dgamma, float* dbeta, mla_stream_t stream) {
    MLA_REQUIRE(x && dy && mean && var && gamma && sums_global && sums_local && count > 0, MLA_E_ARG, "bad bn_bwd_apply arguments");
    MLA_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), MLA_E_ARG, "dgamma and dbeta go together");
    BwdOp op{x, dy, yout, mean, var, ldx, ld_dy, ld_y, act, drop_scale, eps};
    const int64_t total = rows * cols;
    const unsigned grid = dx ? unsigned((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192) : 1u;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (mode == 0)
        hipLaunchKernelGGL(bn_bwd_apply_kernel<0>, dim3(grid), dim3(256), 0, s, op, gamma, rows, int(cols), period, sums_global,
                           sums_local, count, dx, ld_dx, accumulate, dgamma, dbeta);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel<1>, dim3(grid), dim3(256), 0, s, op, gamma, rows, int(cols), period, sums_global,
                           sums_local, count, dx, ld_dx, accumulate, dgamma, dbeta);
    MLA_LAUNCH_OK("bn_bwd_apply");
    return MLA_OK;
}

extern "C" int mla_attention_pool(const float* z, int64_t bags, int T, int K, const float* v_mean, const float* v_var,
                                  const float* v_gamma, const float* v_beta, const float* f_mean, const float* f_var,
                                  const float* f_gamma, const float* f_beta, float eps, float* y, int64_t ldy,
                                  float* att_out, float* cla_out, mla_stream_t stream) {
    MLA_REQUIRE(z && y && v_mean && v_var && v_gamma && v_beta && f_mean && f_var && f_gamma && f_beta, MLA_E_ARG, "null attention_pool argument");
    MLA_REQUIRE(T >= 1 && T <= 16 && K >= 1 && K <= 16 && ldy >= K, MLA_E_SHAPE, "attention_pool supports T, K <= 16 (got %d, %d)", T, K);
    if (bags == 0) return MLA_OK;
    BnParams nv{v_mean, v_var, v_gamma, v_beta}, nf{f_mean, f_var, f_gamma, f_beta};
    hipLaunchKernelGGL(attention_pool_kernel<16>, dim3(unsigned((bags + 15) / 16)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), z, bags, T, K, nv, nf, eps, y, ldy, att_out, cla_out);
    MLA_LAUNCH_OK("attention_pool");
    return MLA_OK;
}

extern "C" int mla_attention_pool_bwd(const float* dy, int64_t ld_dy, const float* att, const float* cla, int64_t bags, int T,
                                      int K, float* du_v, float* du_f, mla_stream_t stream) {
    MLA_REQUIRE(dy && att && cla && du_v && du_f && ld_dy >= K, MLA_E_ARG, "bad attention_pool_bwd arguments");
    MLA_REQUIRE(T >= 1 && T <= 16 && K >= 1 && K <= 16, MLA_E_SHAPE, "attention_pool_bwd supports T, K <= 16 (got %d, %d)", T, K);
    if (bags == 0) return MLA_OK;
    hipLaunchKernelGGL(attention_pool_bwd_kernel<16>, dim3(unsigned((bags + 15) / 16)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), dy, ld_dy, att, cla, bags, T, K, du_v, du_f);
    MLA_LAUNCH_OK("attention_pool_bwd");
    return MLA_OK;
}

extern "C" int mla_linear_small(const float* a, int64_t lda, const float* w, int64_t ldw, const float* bias, float* out,
                                int64_t ldo, int64_t M, int64_t N, int64_t K, mla_stream_t stream) {
    MLA_REQUIRE(a && w && out && M >= 0 && N > 0 && K > 0 && lda >= K && ldw >= K && ldo >= N, MLA_E_ARG, "bad linear_small arguments");
    if (M == 0) return MLA_OK;
    if (N % 4 == 0 && N >= 64 && K <= 64 && N * K * 4 <= 48 * 1024 && ldo % 4 == 0 && mla::aligned(out, 16) && M * N >= (1 << 16)) {
        const int64_t blocks = (M * (N / 4) + 255) / 256;
        hipLaunchKernelGGL(linear_small_wide_kernel, dim3(unsigned(blocks < 1024 ? blocks : 1024)), dim3(256), size_t(N) * K * 4,
                           static_cast<hipStream_t>(stream), a, lda, w, ldw, bias, out, ldo, M, int(N), int(K));
        MLA_LAUNCH_OK("linear_small (wide)");
        return MLA_OK;
    }
    hipLaunchKernelGGL(linear_small_kernel, dim3(unsigned((M * N + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), a, lda, w, ldw, bias, out, ldo, M, int(N), int(K));
    MLA_LAUNCH_OK("linear_small");
    return MLA_OK;
}

extern "C" int mla_linear_narrow(const float* a, int64_t lda, const float* w, int64_t ldw, const float* bias, float* out,
                                 int64_t ldo, int64_t M, int64_t N, int64_t K, mla_stream_t stream) {
    MLA_REQUIRE(a && w && out && M >= 0 && lda >= K && ldw >= K && ldo >= N, MLA_E_ARG, "bad linear_narrow arguments");
    MLA_REQUIRE(N >= 1 && N <= 16 && K >= 4 && K % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0 && K * N * 4 <= 64 * 1024, MLA_E_SHAPE,
                "linear_narrow: N <= 16, K a multiple of 4 with 16-byte aligned rows (N %lld, K %lld)", (long long)N, (long long)K);
    MLA_REQUIRE(mla::aligned(a, 16) && mla::aligned(w, 16), MLA_E_ARG, "linear_narrow operands must be 16-byte aligned");
    if (M == 0) return MLA_OK;
    const unsigned grid = unsigned((M + 15) / 16 < 2048 ? (M + 15) / 16 : 2048);
    const size_t lds = size_t(K) * N * 4;
    hipStream_t s = static_cast<hipStream_t>(stream);
#define MLA_NARROW_CASE(NN) \
    if (N == NN) { hipLaunchKernelGGL(linear_narrow_kernel<NN>, dim3(grid), dim3(256), lds, s, a, lda, w, ldw, bias, out, ldo, M, int(K)); }
    MLA_NARROW_CASE(1) else MLA_NARROW_CASE(2) else MLA_NARROW_CASE(3) else MLA_NARROW_CASE(4) else MLA_NARROW_CASE(5) else
    MLA_NARROW_CASE(6) else MLA_NARROW_CASE(7) else MLA_NARROW_CASE(8) else MLA_NARROW_CASE(9) else MLA_NARROW_CASE(10) else
    MLA_NARROW_CASE(11) else MLA_NARROW_CASE(12) else MLA_NARROW_CASE(13) else MLA_NARROW_CASE(14) else MLA_NARROW_CASE(15) else
    MLA_NARROW_CASE(16)
#undef MLA_NARROW_CASE
    MLA_LAUNCH_OK("linear_narrow");
    return MLA_OK;
}

extern "C" int mla_linear_small_bwd(const float* a, int64_t lda, const float* w, int64_t ldw, const float* dz, int64_t ldz,
                                    int64_t M, int64_t N, int64_t K, float* da, int64_t ldda, float* dw, float* db,
                                    mla_stream_t stream) {
    MLA_REQUIRE(a && w && dz && da && dw && db && M > 0 && N > 0 && K > 0, MLA_E_ARG, "bad linear_small_bwd arguments");
    const int64_t waves = N * K + N;                    // one wave per dw / db element, four per block
    const int64_t blocks_da = (M * K + 255) / 256, blocks_dw = (waves + 3) / 4;
    hipLaunchKernelGGL(linear_small_bwd_kernel, dim3(unsigned(blocks_da > blocks_dw ? blocks_da : blocks_dw)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), a, lda, w, ldw, dz, ldz, M, int(N), int(K), da, ldda, dw, db);
    MLA_LAUNCH_OK("linear_small_bwd");
    return MLA_OK;
}
