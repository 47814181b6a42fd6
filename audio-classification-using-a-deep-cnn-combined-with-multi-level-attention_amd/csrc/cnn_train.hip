// cnn_train.hip -- backward pieces of the VGGish feature stack for the finetune training step
// (train.py:96-97 set_requires_grad(clf, True), then loss.backward() at train.py:137 reaches the
// CNN). All f32, NHWC. The forward in training mode runs the convolutions WITHOUT the fused
// pool (conv.hip, pool = 0) so that the pre-pool activations exist for the pool / ReLU backward.
//
//   maxpool2x2            nn.MaxPool2d(2, 2) forward on a kept activation
//   relu_pool_bwd         dZ (pre-pool resolution) from the gradient of the pooled (or un-pooled)
//                         output: routed to the first maximum of each 2x2 window (torch's
//                         tie rule) and masked by ReLU (output > 0)
//   conv_wgrad            dW[co][tap][ci] = sum_pixels dZ[p][co] * A[p + tap][ci] as an implicit
//                         GEMM with K = pixels on v_mfma_f32_16x16x4_f32: NHWC puts 16 consecutive
//                         channels of one pixel on 16 lanes, which is exactly the f32 MFMA operand
//                         layout (row = channel, k = pixel) -- no transposed copies. Workgroup tile
//                         64 co x 64 ci x 9 taps (144 accumulator registers per lane), the input
//                         patch with halo is staged once and serves all nine taps; split over
//                         images, deterministic two-stage reduction that also restores the
//                         state_dict layout (Cout, Cin, 3, 3).
//   conv1_bwd             Cin = 1 special case: recomputes the four pre-pool outputs of each pooled
//                         pixel (f32 fma chain; the forward's MFMA sums differ in the last bits, which
//                         can only flip the routing between two near-equal maxima), routes the gradient
//                         and reduces dW (64 x 9) and db (64).
#include "common.h"
#include "mma_core.h"

namespace {

using namespace mma;

__global__ __launch_bounds__(256) void maxpool_kernel(const float* __restrict__ a, float* __restrict__ out, int64_t n_out,
                                                      int H, int W, int C) {
    const int c4 = C / 4, WO = W / 2, HO = H / 2;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n_out; i += int64_t(gridDim.x) * 256) {
        const int c = int(i % c4);
        int64_t r = i / c4;
        const int xo = int(r % WO); r /= WO;
        const int yo = int(r % HO);
        const int64_t n = r / HO;
        const f32x4* p = reinterpret_cast<const f32x4*>(a + ((n * H + 2 * yo) * W + 2 * xo) * C) + c;
        const f32x4 v00 = p[0], v01 = p[c4], v10 = p[int64_t(W) * c4], v11 = p[int64_t(W) * c4 + c4];
        f32x4 m;
        m.x = fmaxf(fmaxf(v00.x, v01.x), fmaxf(v10.x, v11.x));
        m.y = fmaxf(fmaxf(v00.y, v01.y), fmaxf(v10.y, v11.y));
        m.z = fmaxf(fmaxf(v00.z, v01.z), fmaxf(v10.z, v11.z));
        m.w = fmaxf(fmaxf(v00.w, v01.w), fmaxf(v10.w, v11.w));
        reinterpret_cast<f32x4*>(out)[i] = m;
    }
}

// pooled: one thread per (n, yo, xo, c): dZ[window] = 0 except the first position equal to the max
// (scan order (0,0),(0,1),(1,0),(1,1)), which receives dP if the max is > 0 (ReLU').
// un-pooled: dZ = dA * (A > 0).
__global__ __launch_bounds__(256) void relu_pool_bwd_kernel(const float* __restrict__ a, const float* __restrict__ d_out,
                                                            float* __restrict__ dz, int64_t total, int H, int W, int C, int pool) {
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < total; i += int64_t(gridDim.x) * 256) {
        if (!pool) {
            dz[i] = a[i] > 0.f ? d_out[i] : 0.f;
            continue;
        }
        const int WO = W / 2, HO = H / 2;
        const int c = int(i % C);
        int64_t r = i / C;
        const int xo = int(r % WO); r /= WO;
        const int yo = int(r % HO);
        const int64_t n = r / HO;
        const int64_t base = ((n * H + 2 * yo) * W + 2 * xo) * C + c;
        const int64_t off[4] = {0, C, int64_t(W) * C, int64_t(W) * C + C};
        float best = a[base];
        int arg = 0;
        _Pragma("unroll") for (int k = 1; k < 4; ++k) {
            const float v = a[base + off[k]];
            if (v > best) { best = v; arg = k; }
        }
        const float g = best > 0.f ? d_out[i] : 0.f;
        _Pragma("unroll") for (int k = 0; k < 4; ++k) dz[base + off[k]] = (k == arg) ? g : 0.f;
    }
}

// Same, and the bias gradient db[c] = sum over pixels of dZ[., c] on the way (it used to be a second pass over dZ, the largest
// tensor of the backward pass). The grid-stride step is a multiple of C, so a thread's channel never changes: each thread sums
// its elements in double precision into slot blockIdx * 256 + t; slot s belongs to channel s % C.
__global__ __launch_bounds__(256) void relu_pool_bwd_bias_kernel(const float* __restrict__ a, const float* __restrict__ d_out,
                                                                 float* __restrict__ dz, int64_t total, int H, int W, int C, int pool,
                                                                 double* __restrict__ slots) {
    double acc = 0.0;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < total; i += int64_t(gridDim.x) * 256) {
        if (!pool) {
            const float g = a[i] > 0.f ? d_out[i] : 0.f;
            dz[i] = g;
            acc += g;
            continue;
        }
        const int WO = W / 2, HO = H / 2;
        const int c = int(i % C);
        int64_t r = i / C;
        const int xo = int(r % WO); r /= WO;
        const int yo = int(r % HO);
        const int64_t n = r / HO;
        const int64_t base = ((n * H + 2 * yo) * W + 2 * xo) * C + c;
        const int64_t off[4] = {0, C, int64_t(W) * C, int64_t(W) * C + C};
        float best = a[base];
        int arg = 0;
        _Pragma("unroll") for (int k = 1; k < 4; ++k) {
            const float v = a[base + off[k]];
            if (v > best) { best = v; arg = k; }
        }
        const float g = best > 0.f ? d_out[i] : 0.f;
        _Pragma("unroll") for (int k = 0; k < 4; ++k) dz[base + off[k]] = (k == arg) ? g : 0.f;
        acc += g;
    }
    slots[int64_t(blockIdx.x) * 256 + threadIdx.x] = acc;
}

// db[c] = sum of the slots of channel c (slots c, c + C, ...), one workgroup per channel, fixed order
__global__ __launch_bounds__(256) void bias_slots_finish_kernel(const double* __restrict__ slots, int64_t n_slots, int C,
                                                                float* __restrict__ db) {
    __shared__ double part[256];
    const int c = blockIdx.x;
    double s = 0.0;
    for (int64_t k = int64_t(c) + int64_t(threadIdx.x) * C; k < n_slots; k += int64_t(256) * C) s += slots[k];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (int(threadIdx.x) < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) db[c] = float(part[0]);
}

// ------------------------------------------------------------------------------------ wgrad ---
constexpr int kPix = 80;                 // LDS floats per pixel row: 64 channels + 16 pad (bank spread for k = pixel)

template <int CIN, int COUT, int H, int W>
struct WCfg {
    static constexpr int TH = W == 32 ? 4 : (W == 16 ? 8 : 12);        // rows per staged band
    static constexpr int BANDS = H / TH;
    static constexpr int PW = W + 2, PH = TH + 2;
    static constexpr int Z_FLOATS = TH * W * kPix, A_FLOATS = PH * PW * kPix;
    static constexpr int LDS_BYTES = (Z_FLOATS + A_FLOATS) * 4;
    static constexpr int TILES_CO = COUT / 64, TILES_CI = CIN / 64;
    static_assert(H % TH == 0 && COUT % 64 == 0 && CIN % 64 == 0 && LDS_BYTES <= 160 * 1024, "wgrad tiling");
};

template <typename C, int CIN, int COUT, int H, int W>
__global__ __launch_bounds__(256, 1) void wgrad_kernel(const float* __restrict__ dz, const float* __restrict__ ain,
                                                       float* __restrict__ partial, int n_img) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sZ = smem;
    float* sA = smem + C::Z_FLOATS;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wr = wave >> 1, wc = wave & 1;
    const int r = lane & 15, q = lane >> 4;
    const int co0 = (blockIdx.x / C::TILES_CI) * 64, ci0 = (blockIdx.x % C::TILES_CI) * 64;
    const int split = blockIdx.y, splits = gridDim.y;

    f32x4 acc[2][2][9];
    _Pragma("unroll") for (int i = 0; i < 2; ++i)
        _Pragma("unroll") for (int j = 0; j < 2; ++j)
            _Pragma("unroll") for (int k = 0; k < 9; ++k) acc[i][j][k] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int img = split; img < n_img; img += splits) {
        for (int band = 0; band < C::BANDS; ++band) {
            const int y0 = band * C::TH;
            __syncthreads();
            // stage dZ rows y0..y0+TH-1 (64 output channels) and the input patch with halo (64 input channels)
            for (int p = t; p < C::TH * W * 16; p += 256) {
                const int ch = p & 15, pix = p >> 4;
                const int x = pix % W, y = pix / W;
                const f32x4 v = *reinterpret_cast<const f32x4*>(dz + ((size_t(img) * H + y0 + y) * W + x) * COUT + co0 + ch * 4);
                *reinterpret_cast<f32x4*>(sZ + pix * kPix + ch * 4) = v;
            }
            for (int p = t; p < C::PH * C::PW * 16; p += 256) {
                const int ch = p & 15, pix = p >> 4;
                const int xh = pix % C::PW, yh = pix / C::PW;
                const int gy = y0 + yh - 1, gx = xh - 1;
                f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
                if (gy >= 0 && gy < H && gx >= 0 && gx < W)
                    v = *reinterpret_cast<const f32x4*>(ain + ((size_t(img) * H + gy) * W + gx) * CIN + ci0 + ch * 4);
                *reinterpret_cast<f32x4*>(sA + pix * kPix + ch * 4) = v;
            }
            __syncthreads();
            // K loop: 4 consecutive pixels of one row per MFMA step
            for (int y = 0; y < C::TH; ++y) {
                for (int x4 = 0; x4 < W; x4 += 4) {
                    float za[2], ab[2][9];
                    _Pragma("unroll") for (int i = 0; i < 2; ++i)
                        za[i] = sZ[(y * W + x4 + q) * kPix + (wr * 2 + i) * 16 + r];
                    _Pragma("unroll") for (int k = 0; k < 9; ++k)
                        _Pragma("unroll") for (int j = 0; j < 2; ++j)
                            ab[j][k] = sA[((y + k / 3) * C::PW + x4 + q + k % 3) * kPix + (wc * 2 + j) * 16 + r];
                    _Pragma("unroll") for (int k = 0; k < 9; ++k)
                        _Pragma("unroll") for (int i = 0; i < 2; ++i)
                            _Pragma("unroll") for (int j = 0; j < 2; ++j)
                                acc[i][j][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(za[i], ab[j][k], acc[i][j][k], 0, 0, 0);
                }
            }
        }
    }
    // partial[split][co][tap][ci]; C/D layout: col = lane & 15 (ci), row = 4 (lane >> 4) + reg (co)
    float* out = partial + size_t(split) * COUT * 9 * CIN;
    _Pragma("unroll") for (int i = 0; i < 2; ++i)
        _Pragma("unroll") for (int j = 0; j < 2; ++j)
            _Pragma("unroll") for (int k = 0; k < 9; ++k) {
                const float v[4] = {acc[i][j][k].x, acc[i][j][k].y, acc[i][j][k].z, acc[i][j][k].w};
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {
                    const int co = co0 + (wr * 2 + i) * 16 + 4 * q + e, ci = ci0 + (wc * 2 + j) * 16 + r;
                    out[(size_t(co) * 9 + k) * CIN + ci] = v[e];
                }
            }
}

// dW[co][ci][tap] (state_dict layout) = sum over splits of partial[split][co][tap][ci]
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int splits, int cout, int cin,
                                                           float* __restrict__ dw) {
    const int64_t total = int64_t(cout) * cin * 9;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < total; i += int64_t(gridDim.x) * 256) {
        const int tap = int(i % 9);
        const int ci = int((i / 9) % cin);
        const int co = int(i / (int64_t(9) * cin));
        const size_t src = (size_t(co) * 9 + tap) * cin + ci;
        float s = 0.f;
        for (int k = 0; k < splits; ++k) s += partial[size_t(k) * total + src];
        dw[i] = s;
    }
}

template <int CIN, int COUT, int H, int W>
int launch_wgrad(const float* dz, const float* ain, int64_t n, float* partial, int64_t partial_floats, float* dw, hipStream_t s) {
    using C = WCfg<CIN, COUT, H, W>;
    const int tiles = C::TILES_CO * C::TILES_CI;
    int splits = (768 + tiles - 1) / tiles;
    if (splits > n) splits = int(n);
    const int64_t need = int64_t(splits) * COUT * 9 * CIN;
    MLA_REQUIRE(partial_floats >= need, MLA_E_ARG, "wgrad workspace too small: %lld < %lld floats", (long long)partial_floats, (long long)need);
    auto kern = wgrad_kernel<C, CIN, COUT, H, W>;
    MLA_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
    hipLaunchKernelGGL(kern, dim3(tiles, splits), dim3(256), C::LDS_BYTES, s, dz, ain, partial, int(n));
    MLA_LAUNCH_OK("wgrad_kernel");
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(1024), dim3(256), 0, s, partial, splits, COUT, CIN, dw);
    MLA_LAUNCH_OK("wgrad_reduce_kernel");
    return MLA_OK;
}

// ----------------------------------------------------------------------------- conv1 backward ---
// block (x = pixel block, y = channel group of 8): each lane owns pooled pixels, recomputes the four
// pre-pool outputs per channel, routes d_pooled to the first maximum and accumulates
// dW (8 x 9) and db (8) in registers; block tree-reduction -> partial[blockIdx.x][cg][80].
__global__ __launch_bounds__(256) void conv1_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, const float* __restrict__ d_pooled,
                                                        int64_t n_pix, float* __restrict__ partial) {
    __shared__ float red[4][80];
    const int cg = blockIdx.y;
    float gw[8][9], gb[8];
    _Pragma("unroll") for (int c = 0; c < 8; ++c) {
        gb[c] = 0.f;
        _Pragma("unroll") for (int k = 0; k < 9; ++k) gw[c][k] = 0.f;
    }
    for (int64_t idx = int64_t(blockIdx.x) * 256 + threadIdx.x; idx < n_pix; idx += int64_t(gridDim.x) * 256) {
        const int px = int(idx & 31), py = int((idx >> 5) % 48);
        const int64_t n = idx / (48 * 32);
        float patch[4][4];
        _Pragma("unroll") for (int a = 0; a < 4; ++a)
            _Pragma("unroll") for (int b = 0; b < 4; ++b) {
                const int iy = 2 * py - 1 + a, ix = 2 * px - 1 + b;
                patch[a][b] = (iy >= 0 && iy < 96 && ix >= 0 && ix < 64) ? x[(n * 96 + iy) * 64 + ix] : 0.f;
            }
        _Pragma("unroll") for (int c = 0; c < 8; ++c) {
            const int ch = cg * 8 + c;
            float o[4] = {0.f, 0.f, 0.f, 0.f};
            _Pragma("unroll") for (int ky = 0; ky < 3; ++ky)
                _Pragma("unroll") for (int kx = 0; kx < 3; ++kx) {
                    const float wv = w[ch * 9 + ky * 3 + kx];
                    o[0] = fmaf(patch[ky][kx], wv, o[0]);
                    o[1] = fmaf(patch[ky][kx + 1], wv, o[1]);
                    o[2] = fmaf(patch[ky + 1][kx], wv, o[2]);
                    o[3] = fmaf(patch[ky + 1][kx + 1], wv, o[3]);
                }
            float best = o[0];
            int arg = 0;
            _Pragma("unroll") for (int k = 1; k < 4; ++k)
                if (o[k] > best) { best = o[k]; arg = k; }
            const float g = (best + bias[ch] > 0.f) ? d_pooled[idx * 64 + ch] : 0.f;
            gb[c] += g;
            const int dy = arg >> 1, dx = arg & 1;
            _Pragma("unroll") for (int ky = 0; ky < 3; ++ky)
                _Pragma("unroll") for (int kx = 0; kx < 3; ++kx) {
                    // patch[dy + ky][dx + kx] with (dy, dx) in {0,1}^2: select without dynamic indexing
                    const float v = dy ? (dx ? patch[ky + 1][kx + 1] : patch[ky + 1][kx]) : (dx ? patch[ky][kx + 1] : patch[ky][kx]);
                    gw[c][ky * 3 + kx] = fmaf(g, v, gw[c][ky * 3 + kx]);
                }
        }
    }
    // wave reduce, then across the 4 waves
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    _Pragma("unroll") for (int c = 0; c < 8; ++c) {
        _Pragma("unroll") for (int k = 0; k < 10; ++k) {
            float v = k < 9 ? gw[c][k] : gb[c];
            _Pragma("unroll") for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if (lane == 0) red[wave][c * 10 + k] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x < 80)
        partial[(size_t(blockIdx.x) * 8 + cg) * 80 + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// one workgroup per (cg, c, k): the partials of all blocks are summed by 256 lanes and a fixed tree (it used to be one serial
// loop over the 1 024 blocks per output: 0.42 ms of the finetune step)
__global__ __launch_bounds__(256) void conv1_bwd_finish_kernel(const float* __restrict__ partial, int blocks, float* __restrict__ dw, float* __restrict__ db) {
    __shared__ double part[256];
    const int i = blockIdx.x;                                      // 0 .. 639: (cg, c, k)
    const int cg = i / 80, rem = i % 80, c = rem / 10, k = rem % 10;
    double s = 0.0;
    for (int b = threadIdx.x; b < blocks; b += 256) s += partial[(size_t(b) * 8 + cg) * 80 + rem];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (int(threadIdx.x) < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (k < 9) dw[(cg * 8 + c) * 9 + k] = float(part[0]);
        else db[cg * 8 + c] = float(part[0]);
    }
}

}  // namespace

extern "C" int mla_maxpool2x2(const float* a, float* out, int64_t n, int H, int W, int C, mla_stream_t stream) {
    MLA_REQUIRE(a && out && n >= 0 && H % 2 == 0 && W % 2 == 0 && C % 4 == 0, MLA_E_ARG, "bad maxpool arguments");
    const int64_t total = n * (H / 2) * (W / 2) * (C / 4);
    if (total == 0) return MLA_OK;
    const unsigned grid = unsigned((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(maxpool_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), a, out, total, H, W, C);
    MLA_LAUNCH_OK("maxpool");
    return MLA_OK;
}

extern "C" int mla_relu_pool_bwd(const float* a, const float* d_out, float* dz, int64_t n, int H, int W, int C, int pool,
                                 mla_stream_t stream) {
    MLA_REQUIRE(a && d_out && dz && n >= 0, MLA_E_ARG, "bad relu_pool_bwd arguments");
    const int64_t total = pool ? n * (H / 2) * (W / 2) * C : n * H * W * C;
    if (total == 0) return MLA_OK;
    const unsigned grid = unsigned((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(relu_pool_bwd_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), a, d_out, dz, total, H, W, C, pool);
    MLA_LAUNCH_OK("relu_pool_bwd");
    return MLA_OK;
}

constexpr int kBiasGrid = 4096;            // workgroups of relu_pool_bwd_bias_kernel (a multiple of C / 256 for C <= 1024)

extern "C" int64_t mla_relu_pool_bwd_bias_workspace_bytes(void) { return int64_t(kBiasGrid) * 256 * 8; }

extern "C" int mla_relu_pool_bwd_bias(const float* a, const float* d_out, float* dz, int64_t n, int H, int W, int C, int pool,
                                      void* workspace, float* db, mla_stream_t stream) {
    MLA_REQUIRE(a && d_out && dz && workspace && db && n > 0, MLA_E_ARG, "bad relu_pool_bwd_bias arguments");
    MLA_REQUIRE(C > 0 && (kBiasGrid * 256) % C == 0, MLA_E_SHAPE, "channel count %d must divide %d", C, kBiasGrid * 256);
    MLA_REQUIRE(!pool || (H % 2 == 0 && W % 2 == 0), MLA_E_SHAPE, "pooling needs even H, W");
    const int64_t total = pool ? n * (H / 2) * (W / 2) * C : n * H * W * C;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(relu_pool_bwd_bias_kernel, dim3(kBiasGrid), dim3(256), 0, s, a, d_out, dz, total, H, W, C, pool,
                       static_cast<double*>(workspace));
    MLA_LAUNCH_OK("relu_pool_bwd_bias");
    hipLaunchKernelGGL(bias_slots_finish_kernel, dim3(unsigned(C)), dim3(256), 0, s, static_cast<const double*>(workspace),
                       int64_t(kBiasGrid) * 256, C, db);
    MLA_LAUNCH_OK("bias_slots_finish");
    return MLA_OK;
}

extern "C" int64_t mla_conv_wgrad_workspace_floats(void) { return int64_t(768 + 64) * 64 * 64 * 9; }

extern "C" int mla_conv_wgrad(const float* dz, const float* a_in, int64_t n, int H, int W, int cin, int cout, float* workspace,
                              int64_t workspace_floats, float* dw_oihw, mla_stream_t stream) {
    MLA_REQUIRE(dz && a_in && workspace && dw_oihw && n > 0, MLA_E_ARG, "bad wgrad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
#define MLA_WGRAD_CASE(CI, CO, HH, WW) \
    if (cin == CI && cout == CO && H == HH && W == WW) return launch_wgrad<CI, CO, HH, WW>(dz, a_in, n, workspace, workspace_floats, dw_oihw, s);
    MLA_WGRAD_CASE(64, 128, 48, 32)
    MLA_WGRAD_CASE(128, 256, 24, 16)
    MLA_WGRAD_CASE(256, 256, 24, 16)
    MLA_WGRAD_CASE(256, 512, 12, 8)
    MLA_WGRAD_CASE(512, 512, 12, 8)
#undef MLA_WGRAD_CASE
    return mla::fail(MLA_E_SHAPE, "wgrad %dx%d %d->%d is not compiled", H, W, cin, cout);
}

// x (n, 96, 64) f32, d_pooled (n, 48, 32, 64): dw (64, 1, 3, 3), db (64). workspace: 1024 * 8 * 80 floats.
extern "C" int mla_conv1_bwd(const float* x, const float* w, const float* bias, const float* d_pooled, int64_t n, float* workspace,
                             float* dw, float* db, mla_stream_t stream) {
    MLA_REQUIRE(x && w && bias && d_pooled && workspace && dw && db && n > 0, MLA_E_ARG, "bad conv1_bwd arguments");
    const int64_t n_pix = n * 48 * 32;
    const int blocks = int((n_pix + 255) / 256 < 1024 ? (n_pix + 255) / 256 : 1024);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(conv1_bwd_kernel, dim3(blocks, 8), dim3(256), 0, s, x, w, bias, d_pooled, n_pix, workspace);
    MLA_LAUNCH_OK("conv1_bwd");
    hipLaunchKernelGGL(conv1_bwd_finish_kernel, dim3(640), dim3(256), 0, s, workspace, blocks, dw, db);
    MLA_LAUNCH_OK("conv1_bwd_finish");
    return MLA_OK;
}
