// train_kernels.hip -- the remaining pieces of the reference's inner training step
// (train.py:124-138): cross-entropy on the sigmoid scores (train.py:372 nn.CrossEntropyLoss,
// mean reduction, applied on top of model.py:268's sigmoid outputs), helpers for the Linear
// backward (transposes feeding the MFMA GEMM, column sums for bias gradients), and the Adam
// update (train.py:369 optim.Adam: betas 0.9/0.999, eps 1e-8, no weight decay, no amsgrad) over
// one flat parameter buffer.
#include "common.h"

namespace {

// out[c][r] = in[r][c]; 32x32 tiles through LDS (padded against bank conflicts)
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, int64_t ld_in, float* __restrict__ out,
                                                        int64_t ld_out, int64_t rows, int64_t cols) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    const int64_t c0 = int64_t(blockIdx.x) * 32, r0 = int64_t(blockIdx.y) * 32;
    _Pragma("unroll") for (int j = 0; j < 32; j += 8) {
        const int64_t r = r0 + ty + j, c = c0 + tx;
        if (r < rows && c < cols) tile[ty + j][tx] = in[r * ld_in + c];
    }
    __syncthreads();
    _Pragma("unroll") for (int j = 0; j < 32; j += 8) {
        const int64_t c = c0 + ty + j, r = r0 + tx;
        if (r < rows && c < cols) out[c * ld_out + r] = tile[tx][ty + j];
    }
}

// loss = inv_total * sum_b (logsumexp(x_b) - x_b[y_b]);  dx = inv_total * (softmax(x_b) - onehot(y_b))
// one block: deterministic double-precision reduction. inv_total = 1 / (global batch).
// A label outside [0, K) (nn.CrossEntropyLoss raises "Target out of bounds" for it) is never used as an index: the row
// contributes nothing, the loss comes out NaN and n_correct[1] = the number of such labels, which the host layer turns into
// the exception (ops.cross_entropy / train_model) without a per-step synchronisation of its own.
__global__ __launch_bounds__(256) void cross_entropy_kernel(const float* __restrict__ x, int64_t ldx,
                                                            const int64_t* __restrict__ labels, int64_t rows, int K,
                                                            float inv_total, float* __restrict__ loss, float* __restrict__ dx,
                                                            int64_t ld_dx, int* __restrict__ n_correct) {
    __shared__ double part[256];
    __shared__ int hits[256];
    __shared__ int bads[256];
    double acc = 0.0;
    int correct = 0, bad = 0;
    for (int64_t b = threadIdx.x; b < rows; b += 256) {
        const float* row = x + b * ldx;
        float mx = row[0];
        int arg = 0;
        for (int k = 1; k < K; ++k)
            if (row[k] > mx) { mx = row[k]; arg = k; }
        float den = 0.f;
        for (int k = 0; k < K; ++k) den += __expf(row[k] - mx);
        const float lse = mx + __logf(den);
        const int64_t y64 = labels[b];
        const bool ok = y64 >= 0 && y64 < K;
        const int y = ok ? int(y64) : -1;
        bad += !ok;
        if (ok) acc += double(lse - row[y]);
        correct += (arg == y);
        if (dx)
            for (int k = 0; k < K; ++k) dx[b * ld_dx + k] = ok ? (__expf(row[k] - lse) - (k == y ? 1.f : 0.f)) * inv_total : 0.f;
    }
    part[threadIdx.x] = acc;
    hits[threadIdx.x] = correct;
    bads[threadIdx.x] = bad;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            part[threadIdx.x] += part[threadIdx.x + o];
            hits[threadIdx.x] += hits[threadIdx.x + o];
            bads[threadIdx.x] += bads[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *loss = bads[0] ? __builtin_nanf("") : float(part[0] * inv_total);
        if (n_correct) { n_correct[0] = hits[0]; n_correct[1] = bads[0]; }
    }
}

// column sums of a (rows, cols) matrix: block (x = 64-column tile, y = row chunk) -> partials
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, int64_t ldx, int64_t rows, int cols,
                                                             double* __restrict__ partial) {
    __shared__ double part[4][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int64_t per = (rows + gridDim.y - 1) / gridDim.y;
    const int64_t r0 = int64_t(blockIdx.y) * per, r1 = r0 + per < rows ? r0 + per : rows;
    double s = 0.0;
    if (c < cols)
        for (int64_t r = r0 + g; r < r1; r += 4) s += x[r * ldx + c];
    part[g][lane] = s;
    __syncthreads();
    if (g == 0 && c < cols) partial[int64_t(blockIdx.y) * cols + c] = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
}

// the same with 16-byte loads: thread (row group g, column quad q) of block (x = 256-column tile, y = row chunk) adds the four
// columns 4 q .. 4 q + 3 of rows g, g + 4, ... of the chunk; a wave reads 1 KiB of one row per instruction (the scalar form: 256 B)
__global__ __launch_bounds__(256) void colsum_partial4_kernel(const float* __restrict__ x, int64_t ldx, int64_t rows, int cols,
                                                              double* __restrict__ partial) {
    __shared__ double part[4][64][4];
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = (blockIdx.x * 64 + lane) * 4;
    const int64_t per = (rows + gridDim.y - 1) / gridDim.y;
    const int64_t r0 = int64_t(blockIdx.y) * per, r1 = r0 + per < rows ? r0 + per : rows;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (c < cols)
        for (int64_t r = r0 + g; r < r1; r += 4) {
            const f32x4_t v = *reinterpret_cast<const f32x4_t*>(x + r * ldx + c);
            s0 += v.x; s1 += v.y; s2 += v.z; s3 += v.w;
        }
    part[g][lane][0] = s0; part[g][lane][1] = s1; part[g][lane][2] = s2; part[g][lane][3] = s3;
    __syncthreads();
    if (g == 0 && c < cols)
        _Pragma("unroll") for (int e = 0; e < 4; ++e)
            partial[int64_t(blockIdx.y) * cols + c + e] = part[0][lane][e] + part[1][lane][e] + part[2][lane][e] + part[3][lane][e];
}

__global__ void colsum_finish_kernel(const double* __restrict__ partial, int chunks, int cols, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    double s = 0.0;
    for (int k = 0; k < chunks; ++k) s += partial[int64_t(k) * cols + c];
    out[c] = float(s);
}

__global__ void axpy_kernel(float a, const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) y[i] += a * x[i];
}

// torch.optim.Adam single-tensor step (no weight decay, no amsgrad):
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// One update function for both kernel forms, with the fused multiply-adds written out and contraction off: left to the
// compiler, `b1 * m + (1 - b1) * g` fuses around either product, and it chose differently in the two kernels (last-bit
// differences between an eager step and its graph replay).
__device__ __forceinline__ void adam_update(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                            float* __restrict__ v, int64_t i, float b1, float b2, float eps, float step_size,
                                            float inv_sqrt_bc2) {
#pragma clang fp contract(off)
    const float gi = g[i];
    const float mi = fmaf(b1, m[i], (1.f - b1) * gi);
    const float vi = fmaf(b2, v[i], (1.f - b2) * (gi * gi));
    m[i] = mi;
    v[i] = vi;
    const float den = fmaf(sqrtf(vi), inv_sqrt_bc2, eps);
    p[i] = p[i] - step_size * mi / den;
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            int64_t n, float b1, float b2, float eps, float step_size, float inv_sqrt_bc2) {
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x)
        adam_update(p, g, m, v, i, b1, b2, eps, step_size, inv_sqrt_bc2);
}

// The same step with its two step-dependent scalars (lr / (1 - b1^t), 1 / sqrt(1 - b2^t)) read from device memory: a HIP graph
// that holds this launch replays with the values of the moment, which mla_adam_prepare writes (computed on the host exactly as
// mla_adam_step computes them, so both forms update the parameters bit for bit alike).
__global__ void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                int64_t n, float b1, float b2, float eps, const float* __restrict__ scal) {
    const float step_size = scal[0], inv_sqrt_bc2 = scal[1];
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x)
        adam_update(p, g, m, v, i, b1, b2, eps, step_size, inv_sqrt_bc2);
}

__global__ void set2_kernel(float* dst, float a, float b) { dst[0] = a; dst[1] = b; }

__global__ void counter_add_kernel(int64_t* counter, int64_t delta) { *counter += delta; }

// splitmix64 finaliser; the same three lines as weights.py _mix (uint64 wrap-around arithmetic only)
__host__ __device__ inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// 16 mask bytes per thread and store (one dwordx4): HBM-write-bound, 1 B per element
__global__ __launch_bounds__(256) void dropout_mask_kernel(uint8_t* __restrict__ mask, int64_t n, uint64_t key, uint64_t offset,
                                                           uint32_t thresh) {
    const uint64_t gold = 0x9E3779B97F4A7C15ull;
    for (int64_t i0 = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) * 16; i0 < n; i0 += int64_t(gridDim.x) * blockDim.x * 16) {
        uint32_t w[4] = {0, 0, 0, 0};
        _Pragma("unroll") for (int j = 0; j < 16; ++j) {
            const uint64_t z = mix64((offset + uint64_t(i0 + j)) * gold + key);
            w[j >> 2] |= uint32_t(uint32_t(z >> 40) >= thresh) << (8 * (j & 3));
        }
        if (i0 + 16 <= n && (reinterpret_cast<uintptr_t>(mask + i0) & 15) == 0) {
            *reinterpret_cast<uint4*>(mask + i0) = make_uint4(w[0], w[1], w[2], w[3]);
        } else {
            for (int j = 0; j < 16 && i0 + j < n; ++j) mask[i0 + j] = uint8_t(w[j >> 2] >> (8 * (j & 3)));
        }
    }
}

// the same mask with stream_id = stream_base + *counter + 1 read on the device (graph replays draw a fresh mask per step)
__global__ __launch_bounds__(256) void dropout_mask_dev_kernel(uint8_t* __restrict__ mask, int64_t n, uint64_t seed, uint64_t stream_base,
                                                               const int64_t* __restrict__ counter, uint64_t offset, uint32_t thresh) {
    const uint64_t gold = 0x9E3779B97F4A7C15ull;
    const uint64_t key = mix64(seed * gold + stream_base + uint64_t(*counter) + 1ull);
    for (int64_t i0 = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) * 16; i0 < n; i0 += int64_t(gridDim.x) * blockDim.x * 16) {
        uint32_t w[4] = {0, 0, 0, 0};
        _Pragma("unroll") for (int j = 0; j < 16; ++j) {
            const uint64_t z = mix64((offset + uint64_t(i0 + j)) * gold + key);
            w[j >> 2] |= uint32_t(uint32_t(z >> 40) >= thresh) << (8 * (j & 3));
        }
        if (i0 + 16 <= n && (reinterpret_cast<uintptr_t>(mask + i0) & 15) == 0) {
            *reinterpret_cast<uint4*>(mask + i0) = make_uint4(w[0], w[1], w[2], w[3]);
        } else {
            for (int j = 0; j < 16 && i0 + j < n; ++j) mask[i0 + j] = uint8_t(w[j >> 2] >> (8 * (j & 3)));
        }
    }
}

unsigned grid_for(int64_t n) { return unsigned((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192); }

}  // namespace

extern "C" int mla_transpose_f32(const float* in, int64_t ld_in, float* out, int64_t ld_out, int64_t rows, int64_t cols,
                                 mla_stream_t stream) {
    MLA_REQUIRE(in && out && rows > 0 && cols > 0 && ld_in >= cols && ld_out >= rows, MLA_E_ARG, "bad transpose arguments");
    hipLaunchKernelGGL(transpose_kernel, dim3(unsigned((cols + 31) / 32), unsigned((rows + 31) / 32)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), in, ld_in, out, ld_out, rows, cols);
    MLA_LAUNCH_OK("transpose");
    return MLA_OK;
}

extern "C" int mla_cross_entropy(const float* x, int64_t ldx, const int64_t* labels, int64_t rows, int K, float inv_total,
                                 float* loss, float* dx, int64_t ld_dx, int* n_correct, mla_stream_t stream) {
    MLA_REQUIRE(x && labels && loss && rows > 0 && K >= 1 && ldx >= K, MLA_E_ARG, "bad cross_entropy arguments");
    hipLaunchKernelGGL(cross_entropy_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), x, ldx, labels, rows, K,
                       inv_total, loss, dx, ld_dx, n_correct);
    MLA_LAUNCH_OK("cross_entropy");
    return MLA_OK;
}

// workspace: 64 * cols doubles
extern "C" int mla_col_sum(const float* x, int64_t ldx, int64_t rows, int64_t cols, void* workspace, float* out,
                           mla_stream_t stream) {
    MLA_REQUIRE(x && workspace && out && rows > 0 && cols > 0 && ldx >= cols, MLA_E_ARG, "bad col_sum arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    int chunks;
    if (cols % 4 == 0 && ldx % 4 == 0 && mla::aligned(x, 16)) {          // 16-byte loads, more row chunks (<= 64: the workspace)
        chunks = int(rows / 64 < 1 ? 1 : (rows / 64 > 64 ? 64 : rows / 64));
        hipLaunchKernelGGL(colsum_partial4_kernel, dim3(unsigned((cols + 255) / 256), unsigned(chunks)), dim3(256), 0, s, x, ldx, rows,
                           int(cols), static_cast<double*>(workspace));
    } else {
        chunks = int(rows / 64 < 1 ? 1 : (rows / 64 > 64 ? 64 : rows / 64));       // narrow matrices (the attention modules' 10 columns): rows are the parallelism
        hipLaunchKernelGGL(colsum_partial_kernel, dim3(unsigned((cols + 63) / 64), unsigned(chunks)), dim3(256), 0, s, x, ldx, rows,
                           int(cols), static_cast<double*>(workspace));
    }
    MLA_LAUNCH_OK("colsum partial");
    hipLaunchKernelGGL(colsum_finish_kernel, dim3(unsigned((cols + 255) / 256)), dim3(256), 0, s,
                       static_cast<const double*>(workspace), chunks, int(cols), out);
    MLA_LAUNCH_OK("colsum finish");
    return MLA_OK;
}

extern "C" int mla_axpy(float a, const float* x, float* y, int64_t n, mla_stream_t stream) {
    MLA_REQUIRE(x && y && n >= 0, MLA_E_ARG, "bad axpy arguments");
    if (n == 0) return MLA_OK;
    hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), a, x, y, n);
    MLA_LAUNCH_OK("axpy");
    return MLA_OK;
}

extern "C" int mla_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                             float eps, int64_t step, mla_stream_t stream) {
    MLA_REQUIRE(p && g && m && v && n >= 0 && step >= 1, MLA_E_ARG, "bad adam arguments");
    if (n == 0) return MLA_OK;
    const double bc1 = 1.0 - pow(double(beta1), double(step)), bc2 = 1.0 - pow(double(beta2), double(step));
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), p, g, m, v, n, beta1, beta2,
                       eps, float(double(lr) / bc1), float(1.0 / sqrt(bc2)));
    MLA_LAUNCH_OK("adam");
    return MLA_OK;
}

extern "C" int mla_adam_prepare(float* scal2_dev, float lr, float beta1, float beta2, int64_t step, mla_stream_t stream) {
    MLA_REQUIRE(scal2_dev && step >= 1, MLA_E_ARG, "bad adam_prepare arguments");
    const double bc1 = 1.0 - pow(double(beta1), double(step)), bc2 = 1.0 - pow(double(beta2), double(step));
    hipLaunchKernelGGL(set2_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), scal2_dev, float(double(lr) / bc1),
                       float(1.0 / sqrt(bc2)));
    MLA_LAUNCH_OK("adam_prepare");
    return MLA_OK;
}

extern "C" int mla_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, float beta1, float beta2, float eps,
                                 const float* scal2_dev, mla_stream_t stream) {
    MLA_REQUIRE(p && g && m && v && scal2_dev && n >= 0, MLA_E_ARG, "bad adam arguments");
    if (n == 0) return MLA_OK;
    hipLaunchKernelGGL(adam_dev_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), p, g, m, v, n, beta1, beta2,
                       eps, scal2_dev);
    MLA_LAUNCH_OK("adam (scalars on the device)");
    return MLA_OK;
}

extern "C" int mla_counter_add(int64_t* counter_dev, int64_t delta, mla_stream_t stream) {
    MLA_REQUIRE(counter_dev, MLA_E_ARG, "null counter");
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), counter_dev, delta);
    MLA_LAUNCH_OK("counter_add");
    return MLA_OK;
}

extern "C" int mla_dropout_mask_dev(uint8_t* mask, int64_t n, uint64_t seed, uint64_t stream_base, const int64_t* counter_dev,
                                    uint64_t offset, float p_drop, mla_stream_t stream) {
    MLA_REQUIRE(mask && counter_dev && n >= 0 && p_drop >= 0.f && p_drop <= 1.f, MLA_E_ARG, "bad dropout_mask arguments");
    if (n == 0) return MLA_OK;
    const uint32_t thresh = uint32_t(llround(double(p_drop) * double(1 << 24)));
    hipLaunchKernelGGL(dropout_mask_dev_kernel, dim3(grid_for((n + 15) / 16)), dim3(256), 0, static_cast<hipStream_t>(stream), mask, n, seed,
                       stream_base, counter_dev, offset, thresh);
    MLA_LAUNCH_OK("dropout_mask (device stream count)");
    return MLA_OK;
}

extern "C" int mla_dropout_mask(uint8_t* mask, int64_t n, uint64_t seed, uint64_t stream_id, uint64_t offset, float p_drop,
                                mla_stream_t stream) {
    MLA_REQUIRE(mask && n >= 0 && p_drop >= 0.f && p_drop <= 1.f, MLA_E_ARG, "bad dropout_mask arguments");
    if (n == 0) return MLA_OK;
    const uint64_t key = mix64(seed * 0x9E3779B97F4A7C15ull + stream_id);
    const uint32_t thresh = uint32_t(llround(double(p_drop) * double(1 << 24)));
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for((n + 15) / 16)), dim3(256), 0, static_cast<hipStream_t>(stream), mask, n, key,
                       offset, thresh);
    MLA_LAUNCH_OK("dropout_mask");
    return MLA_OK;
}
