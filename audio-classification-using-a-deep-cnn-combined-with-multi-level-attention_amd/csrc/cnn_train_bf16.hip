// cnn_train_bf16.hip -- the finetune backward of the VGGish feature stack in bf16 (train.py:96-97 set_requires_grad(clf, True),
// then loss.backward() at train.py:137 reaches the CNN): bf16 activations and gradients, f32 accumulation, f32 weight
// gradients into the flat gradient buffer, f32 master weights (the bf16 weight copies are re-derived after every Adam step).
// The f32 forms (exact-MFMA parity mode) live in cnn_train.hip; this file is what the step runs in at speed.
//
//   maxpool2x2_bf16       nn.MaxPool2d(2, 2) on a kept pre-pool activation, 8 channels (16 B) per lane
//   relu_pool_bwd_bf16    dZ from the gradient of relu(.) / maxpool(relu(.)): first-maximum routing (torch's tie rule) and
//                         ReLU mask; the layer's bias gradient (column sums of dZ) is accumulated on the way
//   wgrad_bf16            dW[co][tap][ci] = sum_pixels dZ[p][co] * A[p + tap][ci] as an implicit GEMM with K = pixels on
//                         v_mfma_f32_16x16x32_bf16. NHWC keeps a pixel's channels contiguous while the MFMA wants 8
//                         consecutive k (pixels) of ONE channel per lane: the staged [pixel][channel] tiles are read with
//                         ds_read_b64_tr_b16 (gfx950's transposing LDS read: a 16-lane group fetches 4 pixel rows x 16
//                         channels and each lane receives one channel's 4 pixels), so no transposed copy of dZ or of the
//                         activations ever exists. The order of the 32 pixels inside a k-step is chosen so that each
//                         32-lane half reads 8 CONSECUTIVE pixel rows: with a 160-byte row pitch these are 8 distinct
//                         32-byte bank slots -- conflict-free for every tap shift.
//   conv1_bwd (bf16 dY)   recompute and weight gradient of the Cin = 1 layer as two GEMMs on the matrix cores (conv1_bwd_mfma_kernel)
//   transpose_bf16, col_sum_bf16   helpers of the Linear backward (K-contiguous operands for the MFMA GEMM; bias gradients)
//
// Roofline: wgrad is MFMA-bound (2 * pixels * 9 Cin Cout flop per image, the same as the forward layer); the elementwise
// kernels are HBM-bound (they read and write each activation-sized tensor once).
#include "common.h"
#include "mma_core.h"

namespace {

using namespace mma;

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// eight consecutive channels as floats, from bf16 (one 16-byte load) or f32 (two)
template <typename T> __device__ __forceinline__ void load8(const T* p, float* v);
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float* v) {
    const u32x4 u = *reinterpret_cast<const u32x4*>(p);
    _Pragma("unroll") for (int k = 0; k < 4; ++k) {
        v[2 * k] = __builtin_bit_cast(float, u[k] << 16);
        v[2 * k + 1] = __builtin_bit_cast(float, u[k] & 0xffff0000u);
    }
}
template <> __device__ __forceinline__ void load8<float>(const float* p, float* v) {
    const f32x4 a = reinterpret_cast<const f32x4*>(p)[0], b = reinterpret_cast<const f32x4*>(p)[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void store8(bf16_t* p, const float* v) {
    *reinterpret_cast<u32x4*>(p) = u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
}

__global__ __launch_bounds__(256) void maxpool_bf16_kernel(const bf16_t* __restrict__ a, bf16_t* __restrict__ out, int64_t n_out,
                                                           int H, int W, int C) {
    const int c8 = C / 8, WO = W / 2, HO = H / 2;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n_out; i += int64_t(gridDim.x) * 256) {
        const int c = int(i % c8);
        int64_t r = i / c8;
        const int xo = int(r % WO); r /= WO;
        const int yo = int(r % HO);
        const int64_t n = r / HO;
        const bf16_t* p = a + ((n * H + 2 * yo) * W + 2 * xo) * C + c * 8;
        float v00[8], v01[8], v10[8], v11[8], m[8];
        load8(p, v00); load8(p + C, v01); load8(p + int64_t(W) * C, v10); load8(p + int64_t(W) * C + C, v11);
        _Pragma("unroll") for (int k = 0; k < 8; ++k) m[k] = fmaxf(fmaxf(v00[k], v01[k]), fmaxf(v10[k], v11[k]));
        store8(out + i * 8, m);                              // max of bf16 values is a bf16 value: the repack is exact
    }
}

// One lane per (n, yo, xo, 8 channels) [pool] or per 8 consecutive elements [no pool]; dZ in bf16. The grid-stride step is
// a multiple of C / 8, so a lane's 8 channels never change: their dZ sums go to slots[(block * 256 + t) * 8 + k] in double
// precision and bias_slots_finish8 adds the slots of a channel in fixed order (deterministic).
template <typename TA, typename TD>
__global__ __launch_bounds__(256) void relu_pool_bwd_bf16_kernel(const TA* __restrict__ a, const TD* __restrict__ d_out,
                                                                 bf16_t* __restrict__ dz, int64_t total8, int H, int W, int C, int pool,
                                                                 double* __restrict__ slots) {
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int c8 = C / 8;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < total8; i += int64_t(gridDim.x) * 256) {
        float g[8], d[8];
        load8(d_out + i * 8, d);
        if (!pool) {
            float av[8];
            load8(a + i * 8, av);
            _Pragma("unroll") for (int k = 0; k < 8; ++k) g[k] = av[k] > 0.f ? d[k] : 0.f;
            store8(dz + i * 8, g);
        } else {
            const int WO = W / 2, HO = H / 2;
            const int c = int(i % c8);
            int64_t r = i / c8;
            const int xo = int(r % WO); r /= WO;
            const int yo = int(r % HO);
            const int64_t n = r / HO;
            const int64_t base = ((n * H + 2 * yo) * W + 2 * xo) * C + c * 8;
            const int64_t off[4] = {0, C, int64_t(W) * C, int64_t(W) * C + C};
            float v[4][8];
            _Pragma("unroll") for (int w = 0; w < 4; ++w) load8(a + base + off[w], v[w]);
            int arg[8];
            _Pragma("unroll") for (int k = 0; k < 8; ++k) {
                float best = v[0][k];
                arg[k] = 0;
                _Pragma("unroll") for (int w = 1; w < 4; ++w)
                    if (v[w][k] > best) { best = v[w][k]; arg[k] = w; }
                g[k] = best > 0.f ? d[k] : 0.f;
            }
            _Pragma("unroll") for (int w = 0; w < 4; ++w) {
                float o[8];
                _Pragma("unroll") for (int k = 0; k < 8; ++k) o[k] = arg[k] == w ? g[k] : 0.f;
                store8(dz + base + off[w], o);
            }
        }
        _Pragma("unroll") for (int k = 0; k < 8; ++k) acc[k] += double(g[k]);
    }
    if (slots) {
        double* s = slots + (int64_t(blockIdx.x) * 256 + threadIdx.x) * 8;
        _Pragma("unroll") for (int k = 0; k < 8; ++k) s[k] = acc[k];
    }
}

// db[c]: lane t of block b holds channels 8 ((b * 256 + t) % (C / 8)) + k
__global__ __launch_bounds__(256) void bias_slots_finish8_kernel(const double* __restrict__ slots, int64_t n_lanes, int C,
                                                                 float* __restrict__ db) {
    __shared__ double part[256];
    const int c = blockIdx.x, c8 = C / 8, grp = c / 8, k = c % 8;
    double s = 0.0;
    for (int64_t l = int64_t(grp) + int64_t(threadIdx.x) * c8; l < n_lanes; l += int64_t(256) * c8) s += slots[l * 8 + k];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (int(threadIdx.x) < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) db[c] = float(part[0]);
}

// Pool / ReLU backward from the window CODES the training forward wrote (mla_conv3x3_train_codes: one byte per pooled element, the
// position of the first maximum or 4 where the ReLU is off) instead of the pre-pool activation: 1 + 2 bytes read per pooled element
// instead of 8 + 2, same dZ. One lane per (n, yo, xo, 8 channels); bias slots as relu_pool_bwd_bf16_kernel.
__global__ __launch_bounds__(256) void pool_bwd_codes_bf16_kernel(const uint8_t* __restrict__ codes, const bf16_t* __restrict__ d_out,
                                                                  bf16_t* __restrict__ dz, int64_t total8, int H, int W, int C,
                                                                  double* __restrict__ slots) {
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int c8 = C / 8, WO = W / 2, HO = H / 2;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < total8; i += int64_t(gridDim.x) * 256) {
        float d[8], g[8];
        load8(d_out + i * 8, d);
        const u32x2 cw = *reinterpret_cast<const u32x2*>(codes + i * 8);
        const int c = int(i % c8);
        int64_t r = i / c8;
        const int xo = int(r % WO); r /= WO;
        const int yo = int(r % HO);
        const int64_t n = r / HO;
        const int64_t base = ((n * H + 2 * yo) * W + 2 * xo) * C + c * 8;
        const int64_t off[4] = {0, C, int64_t(W) * C, int64_t(W) * C + C};
        uint32_t code[8];
        _Pragma("unroll") for (int k = 0; k < 8; ++k) {
            code[k] = (cw[k >> 2] >> (8 * (k & 3))) & 0xffu;
            g[k] = code[k] < 4u ? d[k] : 0.f;
            acc[k] += double(g[k]);
        }
        _Pragma("unroll") for (int w = 0; w < 4; ++w) {
            float o[8];
            _Pragma("unroll") for (int k = 0; k < 8; ++k) o[k] = code[k] == uint32_t(w) ? g[k] : 0.f;
            store8(dz + base + off[w], o);
        }
    }
    if (slots) {
        double* s = slots + (int64_t(blockIdx.x) * 256 + threadIdx.x) * 8;
        _Pragma("unroll") for (int k = 0; k < 8; ++k) s[k] = acc[k];
    }
}

// ------------------------------------------------------------------------------------ wgrad ---
// One persistent 8-wave workgroup per CU; waves as WCO x WCI, each 32 co x 32 ci x 9 taps (144 accumulator registers); workgroup
// tile (32 WCO) co x (32 WCI) ci. Work items = (image, band of TH rows) of this workgroup's image split. Per item the dZ band and
// the input patch with halo are staged by LDS-DMA (buffer_load ... lds) into one of TWO LDS images, so the transfer of item i+1 runs
// under the MFMAs of item i (the first version staged synchronously through registers: 600 TFLOP/s; this one: see DESIGN.md).
// LDS image: per 64-channel block, [pixel][128 B] rows (patch rows pitched at a multiple of 8 pixels), 16-byte chunk c of row r
// stored at chunk c ^ (((r >> 1) & 3) << 1): the transposing reads of a 32-lane half cover 8 CONSECUTIVE pixel rows x 32 B, which
// this XOR spreads over the 8 distinct 32-byte slots of a 256-byte bank row pair for every tap shift; because every (k-step, tap
// row) offset is a multiple of 8 rows the key depends on the lane and the tap's kx only -- 8 address registers per lane, all other
// offsets are immediates. The DMA applies the same involution on the source side (a piece = 8 rows x 8 chunks in lane order). Out-of-image halo pixels carry an out-of-range offset: the range check drops those lanes, so the
// x-halo columns are zeroed once and the y-halo rows whenever a first / last band is staged.
#ifndef MLA_WGRAD_COT
#define MLA_WGRAD_COT 4
#endif
#ifndef MLA_WGRAD_DMA_STAGGER
#define MLA_WGRAD_DMA_STAGGER 1
#endif
#ifndef MLA_WGRAD_TCO128
#define MLA_WGRAD_TCO128 1              // workgroup tile 128 co x 64 ci for every shape (0: 64 x 128 where Cin >= 128): the haloed, pitch-padded input
#endif                                // patch is the expensive operand to stage (1.9x its useful bytes), the dZ band is not: conv3 / conv4 +5 ... 6 %
template <int CIN, int COUT, int H, int W>
struct WBCfg {
    // per-wave tile: COT x CIT 16-channel tiles x 9 taps (COT * CIT * 9 = 36 accumulator tiles). 4 x 1 (64 co x 16 ci) instead of
    // 2 x 2: a dZ fragment then feeds 9 MFMAs and an input fragment 4 (2 x 2: 18 and 2), i.e. 26 instead of 40 transposing LDS
    // reads per 36 MFMAs -- at 2 x 2 the eight waves asked the LDS array for ~139 B/clk, more than its 128
    static constexpr int COT = MLA_WGRAD_COT, CIT = 4 / COT;
    static constexpr int TCO = (MLA_WGRAD_TCO128 || CIN < 128) ? 128 : 64, TCI = (MLA_WGRAD_TCO128 || CIN < 128) ? 64 : 128;     // workgroup tile
    static constexpr int WCO = TCO / (16 * COT), WCI = TCI / (16 * CIT);
    static_assert(WCO * WCI == 8, "eight waves");
    static constexpr int CBZ = TCO / 64, CBA = TCI / 64;              // 64-channel blocks per image
    static constexpr int TH = W == 32 ? 4 : (W == 16 ? 8 : 12);        // image rows per staged band
    static constexpr int KSTEPS = TH * W / 32;                         // 32 pixels per MFMA k-step
    static constexpr bool UNROLL = !(W == 32 || (CIN == 512 && COUT == 512));
    // per shape like UNROLL: measured +7 / +1 / +4 / +7 % for conv2 / conv3 / conv4 / conv6; the 256 -> 512 shape spills 24 registers with it
    // (and a spill reload waits for every outstanding DMA: -59 %)
    static constexpr bool DMA_STAGGER = MLA_WGRAD_DMA_STAGGER && !(CIN == 256 && COUT == 512);
    static constexpr int BANDS = H / TH;
    static constexpr int PW = W + 2, PH = TH + 2;
    static constexpr int PWP = (PW + 7) / 8 * 8;                       // patch row pitch in pixel rows: a multiple of 8, so that the
                                                                       // swizzle key of a pixel depends on its column only
    static constexpr int Z_PIX = TH * W, A_PIX = PH * PWP;             // pixel rows per block image (whole DMA pieces)
    static constexpr int Z_BYTES = CBZ * Z_PIX * 128, A_BYTES = CBA * A_PIX * 128;
    static constexpr int BUF_BYTES = Z_BYTES + A_BYTES;
    static constexpr int LDS_BYTES = 2 * BUF_BYTES;
    static constexpr int TILES_CO = COUT / TCO, TILES_CI = CIN / TCI;
    static_assert(H % TH == 0 && Z_PIX % 32 == 0 && COUT % TCO == 0 && CIN % TCI == 0 && LDS_BYTES <= 160 * 1024, "wgrad tiling");
};

__device__ __forceinline__ s16x4 tr_read(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}
__device__ __forceinline__ bf16x8 frag_of(s16x4 lo, s16x4 hi) {
    return __builtin_bit_cast(bf16x8, s16x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w});
}
template <typename C, int CIN, int COUT, int H, int W>
__global__ __launch_bounds__(512, 2) void wgrad_bf16_kernel(const bf16_t* __restrict__ dz, const bf16_t* __restrict__ ain,
                                                            float* __restrict__ partial, int n_img) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wr = wave / C::WCI, wc = wave % C::WCI;
    const int g = lane >> 4, li = lane & 15, q4 = li >> 2, p4 = li & 3;      // transposing read: lane 4 q4 + p4 of its group
    const int r = lane & 15, q = lane >> 4;                                  // supplies pixel row q4, channels 4 p4 .. 4 p4 + 3
    // blockIdx.x = image split (fastest: consecutive workgroups are dealt round-robin over the 8 XCDs, so all tiles of one split --
    // which read the SAME images -- share an XCD's L2 and each image leaves HBM once, not once per XCD), blockIdx.y = (co, ci) tile
    const int co0 = (blockIdx.y / C::TILES_CI) * C::TCO, ci0 = (blockIdx.y % C::TILES_CI) * C::TCI;
    const int split = blockIdx.x, splits = gridDim.x;

    // pixel (y, x) of k-step s, half h (elements 4h .. 4h+3 of the fragment), lane group g, row q4 of the 4-pixel block:
    //   W = 32: y = s,               x = 16 h + 4 g + q4      W = 16: y = 2 s + h, x = 4 g + q4
    //   W = 8 : y = 4 s + 2 h + (g >> 1), x = 4 (g & 1) + q4
    const int ly = W == 8 ? (g >> 1) : 0;
    const int lx = (W == 8 ? 4 * (g & 1) : 4 * g) + q4;
    auto sh_y = [](int s, int h) { return W == 32 ? s : (W == 16 ? 2 * s + h : 4 * s + 2 * h); };
    auto sh_x = [](int h) { return W == 32 ? 16 * h : 0; };
    // this wave's 32 co = 16-channel tiles 2 wr, 2 wr + 1 of the workgroup tile (ci: 2 wc, 2 wc + 1): 64-channel block and 32-byte slot
    // of each. Per-lane byte offsets of the transposing reads; everything that depends on (k-step, half, ky) is a multiple of 8 rows.
    int zbase[C::COT], abase[3][C::CIT];
    _Pragma("unroll") for (int i = 0; i < C::COT; ++i) {
        const int tile = C::COT * wr + i, chunk = 2 * (tile & 3) + (p4 >> 1), key = (lx >> 1) & 3;
        zbase[i] = (tile >> 2) * C::Z_PIX * 128 + (ly * W + lx) * 128 + 16 * (chunk ^ (key << 1)) + 8 * (p4 & 1);
    }
    _Pragma("unroll") for (int kx = 0; kx < 3; ++kx)
        _Pragma("unroll") for (int j = 0; j < C::CIT; ++j) {
            const int tile = C::CIT * wc + j, chunk = 2 * (tile & 3) + (p4 >> 1), key = ((lx + kx) >> 1) & 3;
            abase[kx][j] = (tile >> 2) * C::A_PIX * 128 + (ly * C::PWP + lx + kx) * 128 + 16 * (chunk ^ (key << 1)) + 8 * (p4 & 1);
        }

    constexpr uint32_t IMG_Z = uint32_t(H) * W * COUT * 2, IMG_A = uint32_t(H) * W * CIN * 2;
    // DMA of one item into buffer `buf`: pieces of 8 pixel rows x 128 B, wave `wave` takes pieces wave, wave + 8, ...
    auto stage = [&](int img, int band, int buf) {
        char* base = smem + buf * C::BUF_BYTES;
        const int y0 = band * C::TH;
        const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(dz) + size_t(img) * H * W * COUT, 0, IMG_Z, 0x00020000);
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(ain) + size_t(img) * H * W * CIN, 0, IMG_A, 0x00020000);
        const int prow = lane >> 3, slot = lane & 7;
        constexpr int ZP = C::CBZ * C::Z_PIX / 8, AP = C::CBA * C::A_PIX / 8;
        _Pragma("unroll") for (int p = 0; p < (ZP + 7) / 8; ++p) {
            const int piece = wave + 8 * p;
            if (piece < ZP) {
                const int blk = piece / (C::Z_PIX / 8), row = (piece % (C::Z_PIX / 8)) * 8 + prow;
                const int chunk = slot ^ (((row >> 1) & 3) << 1);
                const int voff = ((y0 * W + row) * COUT + co0 + blk * 64 + chunk * 8) * 2;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rz, (__attribute__((address_space(3))) void*)(base + piece * 1024), 16, voff, 0, 0, 0);
            }
        }
        char* pbase = base + C::Z_BYTES;
        _Pragma("unroll") for (int p = 0; p < (AP + 7) / 8; ++p) {
            const int piece = wave + 8 * p;
            if (piece < AP) {
                const int blk = piece / (C::A_PIX / 8), row = (piece % (C::A_PIX / 8)) * 8 + prow;
                const int xh = row % C::PWP, yh = row / C::PWP;
                const int gy = y0 + yh - 1, gx = xh - 1;
                const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;               // also false for the pitch padding (xh >= PW)
                const int chunk = slot ^ (((row >> 1) & 3) << 1);
                const int voff = ok ? ((gy * W + gx) * CIN + ci0 + blk * 64 + chunk * 8) * 2 : 0x7fffff00;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(pbase + piece * 1024), 16, voff, 0, 0, 0);
            }
        }
        // out-of-image y-halo rows of a first / last band: the DMA dropped them, a previous item may have left data there
        if (C::BANDS > 1 && (band == 0 || band == C::BANDS - 1)) {
            const int yh = band == 0 ? 0 : C::PH - 1;
            for (int i = t; i < C::CBA * C::PWP * 8; i += 512) {
                const int blk = i / (C::PWP * 8), rem = i % (C::PWP * 8);
                *reinterpret_cast<u32x4*>(pbase + (blk * C::A_PIX + yh * C::PWP + rem / 8) * 128 + (rem & 7) * 16) = zero16();
            }
        }
    };

    f32x4 acc[C::COT][C::CIT][9];
    _Pragma("unroll") for (int i = 0; i < C::COT; ++i)
        _Pragma("unroll") for (int j = 0; j < C::CIT; ++j)
            _Pragma("unroll") for (int k = 0; k < 9; ++k) acc[i][j][k] = f32x4{0.f, 0.f, 0.f, 0.f};

    // zero both patch images once: the x-halo columns (and, for single-band shapes, the y-halo rows) are never written by the DMA
    for (int i = t; i < 2 * C::A_BYTES / 16; i += 512) {
        const int buf = i / (C::A_BYTES / 16), off = i % (C::A_BYTES / 16);
        *reinterpret_cast<u32x4*>(smem + buf * C::BUF_BYTES + C::Z_BYTES + off * 16) = zero16();
    }
    __syncthreads();

    const int n_mine = split < n_img ? (n_img - split + splits - 1) / splits : 0;
    const int n_items = n_mine * C::BANDS;
    if (n_items > 0) stage(split, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int it = 0; it < n_items; ++it) {
        const int buf = it & 1;
        // the next item lands under this item's MFMAs. Issuing an item's ~10 DMA pieces stalls a wave for up to a thousand cycles
        // (gemm.hip's stamps); with MLA_WGRAD_DMA_STAGGER the two waves of a SIMD do not do that at the same time: waves 0-3 issue in
        // front of the first k-step, waves 4-7 behind it
        const int ni = it + 1;
        const bool more = ni < n_items;
        const bool issue_late = C::DMA_STAGGER && __builtin_amdgcn_readfirstlane(wave) >= 4;
        if (more && !issue_late) stage(split + (ni / C::BANDS) * splits, ni % C::BANDS, buf ^ 1);
        const char* sZ = smem + buf * C::BUF_BYTES;
        const char* sA = sZ + C::Z_BYTES;
        // k-step body. Whether the loop over the band's k-steps is unrolled is a per-shape choice (C::UNROLL): unrolled, every LDS
        // offset is an immediate but hipcc keeps more addresses live (spills at W = 32 and for 512 -> 512: 611 -> 941 and
        // 981 -> 1200 TFLOP/s when NOT unrolled); the other three shapes lose 7-11 % without it
        auto kstep = [&](int s) {
            bf16x8 za[C::COT];
            _Pragma("unroll") for (int i = 0; i < C::COT; ++i) {
                s16x4 part[2];
                _Pragma("unroll") for (int h = 0; h < 2; ++h)
                    part[h] = tr_read(sZ + zbase[i] + (sh_y(s, h) * W + sh_x(h)) * 128);
                za[i] = frag_of(part[0], part[1]);
            }
            // (reading every fragment of a k-step first and issuing its 36 MFMAs as one burst, as conv.hip does, measured 7-8 % SLOWER on the
            // unrolled shapes: hipcc's own interleaving across the unrolled k-steps is the better schedule here)
            _Pragma("unroll") for (int ky = 0; ky < 3; ++ky) {
                bf16x8 ab[3][C::CIT];
                _Pragma("unroll") for (int kx = 0; kx < 3; ++kx)
                    _Pragma("unroll") for (int j = 0; j < C::CIT; ++j) {
                        s16x4 part[2];
                        _Pragma("unroll") for (int h = 0; h < 2; ++h)
                            part[h] = tr_read(sA + abase[kx][j] + ((sh_y(s, h) + ky) * C::PWP + sh_x(h)) * 128);
                        ab[kx][j] = frag_of(part[0], part[1]);
                    }
                _Pragma("unroll") for (int kx = 0; kx < 3; ++kx)
                    _Pragma("unroll") for (int i = 0; i < C::COT; ++i)
                        _Pragma("unroll") for (int j = 0; j < C::CIT; ++j)
                            acc[i][j][ky * 3 + kx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(za[i], ab[kx][j], acc[i][j][ky * 3 + kx], 0, 0, 0);
            }
        };
        constexpr int S0 = C::DMA_STAGGER ? 1 : 0;             // the first k-step is peeled only where the late issue goes behind it
        if constexpr (C::DMA_STAGGER) {
            kstep(0);
            if (more && issue_late) stage(split + (ni / C::BANDS) * splits, ni % C::BANDS, buf ^ 1);
        }
        if constexpr (C::UNROLL) {
            _Pragma("unroll") for (int s = S0; s < C::KSTEPS; ++s) kstep(s);
        } else {
            _Pragma("unroll 1") for (int s = S0; s < C::KSTEPS; ++s) kstep(s);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the next item has landed before the barrier publishes it
        __syncthreads();                                       // ... and everybody is done reading this one
    }
    // partial[split][co][tap][ci]; C/D layout: col = lane & 15 (ci), row = 4 (lane >> 4) + reg (co)
    float* out = partial + size_t(split) * COUT * 9 * CIN;
    _Pragma("unroll") for (int i = 0; i < C::COT; ++i)
        _Pragma("unroll") for (int j = 0; j < C::CIT; ++j)
            _Pragma("unroll") for (int k = 0; k < 9; ++k) {
                const float v[4] = {acc[i][j][k].x, acc[i][j][k].y, acc[i][j][k].z, acc[i][j][k].w};
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {
                    const int co = co0 + (wr * C::COT + i) * 16 + 4 * q + e, ci = ci0 + (wc * C::CIT + j) * 16 + r;
                    out[(size_t(co) * 9 + k) * CIN + ci] = v[e];
                }
            }
}

// dW[co][ci][tap] (state_dict layout) = sum over splits of partial[split][co][tap][ci]
__global__ __launch_bounds__(256) void wgrad_reduce_bf16_kernel(const float* __restrict__ partial, int splits, int cout, int cin,
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
int launch_wgrad_bf16(const bf16_t* dz, const bf16_t* ain, int64_t n, float* partial, int64_t partial_floats, float* dw, hipStream_t s) {
    using C = WBCfg<CIN, COUT, H, W>;
    const int tiles = C::TILES_CO * C::TILES_CI;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    int splits = cus / tiles < 1 ? 1 : cus / tiles;           // one persistent workgroup per CU
    if (splits > n) splits = int(n);
    const int64_t need = int64_t(splits) * COUT * 9 * CIN;
    MLA_REQUIRE(partial_floats >= need, MLA_E_ARG, "wgrad workspace too small: %lld < %lld floats", (long long)partial_floats, (long long)need);
    auto kern = wgrad_bf16_kernel<C, CIN, COUT, H, W>;
    MLA_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
    hipLaunchKernelGGL(kern, dim3(splits, tiles), dim3(512), C::LDS_BYTES, s, dz, ain, partial, int(n));
    MLA_LAUNCH_OK("wgrad_bf16_kernel");
    hipLaunchKernelGGL(wgrad_reduce_bf16_kernel, dim3(1024), dim3(256), 0, s, partial, splits, COUT, CIN, dw);
    MLA_LAUNCH_OK("wgrad_reduce_bf16_kernel");
    return MLA_OK;
}

// ----------------------------------------------------------------------------- conv1 backward ---
// as conv1_bwd_kernel of cnn_train.hip, the incoming gradient in bf16
__global__ __launch_bounds__(256) void conv1_bwd_bf16_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, const bf16_t* __restrict__ d_pooled,
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
        float dv[8];
        load8(d_pooled + idx * 64 + cg * 8, dv);
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
            const float gg = (best + bias[ch] > 0.f) ? dv[c] : 0.f;
            gb[c] += gg;
            const int dy = arg >> 1, dx = arg & 1;
            _Pragma("unroll") for (int ky = 0; ky < 3; ++ky)
                _Pragma("unroll") for (int kx = 0; kx < 3; ++kx) {
                    const float v = dy ? (dx ? patch[ky + 1][kx + 1] : patch[ky + 1][kx]) : (dx ? patch[ky][kx + 1] : patch[ky][kx]);
                    gw[c][ky * 3 + kx] = fmaf(gg, v, gw[c][ky * 3 + kx]);
                }
        }
    }
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
__global__ __launch_bounds__(256) void conv1_bwd_finish_bf16_kernel(const float* __restrict__ partial, int blocks, float* __restrict__ dw, float* __restrict__ db) {
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

// ---- conv1 backward on the matrix cores (round 2) ------------------------------------------------------------------------------
// conv1 + ReLU + 2x2 max-pool as the patch GEMM of conv.hip (conv1_patch_kernel): pre[pos][c][P] = sum_k W'[pos][c][k] patch[k][P]
// over the 4 x 4 input patch of pooled pixel P. Its weight gradient is a GEMM too: with g[P][c] = dY[P][c] where the pooled output
// is positive and G_pos = g masked to the pixels whose first maximum sits at window position pos,
//     dW'[pos] (64 x 16) = G_pos (64 x pixels) . patch^T (pixels x 16),     dW[c][ky][kx] = sum_pos dW'[pos][c][4 (dy + ky) + dx + kx].
// One wave per pooled row (32 pixels): (1) RECOMPUTE pre^T[P][c] = patch^T W'^T on v_mfma_f32_16x16x32_bf16 (K = 16 patch elements,
// zero-padded) with the same bf16 operands as the forward kernel; in the C/D layout a lane then holds ONE channel's values of 4
// consecutive pixels, for each of the 4 positions -- arg-max, ReLU mask and the gradient (fetched with the transposing LDS read,
// which delivers exactly one channel's 4 pixels per lane) are lane-local; (2) the masked gradients of two 16-pixel tiles ARE the A
// fragment of the second GEMM (row = channel, k = 8 pixels) if its K index enumerates the row's pixels as {4q .. 4q+3, 16+4q ..
// 16+4q+3} for lane group q -- K is a summation index, any order serves as long as the B operand (patch values) uses the same one.
// No transpose, no per-lane 72-register accumulator file, 48 MFMAs per 32 pixels x 64 channels instead of ~82 vector instructions per
// (pixel, channel). Deterministic: static row shares, waves of a workgroup added in order, workgroups summed in a fixed tree.
#ifndef MLA_CONV1_BWD_MFMA
#define MLA_CONV1_BWD_MFMA 1
#endif
constexpr int kC1Part = 4 * 64 * 16 + 64;                   // floats per workgroup partial: dW'[pos][c][k], db[c]
constexpr int kC1MaxWg = 768;                               // persistent workgroups (3 per CU: 43 KB of LDS each)

__global__ __launch_bounds__(256, 2) void conv1_bwd_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, const bf16_t* __restrict__ d_pooled,
                                                             int n_seg, float* __restrict__ partial) {
    constexpr int XP = 72;                                      // bf16 per staged input row: element (gy, gx) at column gx + 1
    __shared__ __attribute__((aligned(16))) uint16_t sW[4 * 64 * 16];
    __shared__ __attribute__((aligned(16))) uint16_t sX[4][4 * XP];
    __shared__ __attribute__((aligned(16))) uint16_t sD[4][32 * 64];
    __shared__ float sRed[kC1Part];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 15, q = lane >> 4, q4 = r >> 2, p4 = r & 3;

    for (int i = t; i < 4 * 64 * 16; i += 256) {                // W'[pos][c][k]: the 3 x 3 filter at offset (dy, dx) inside the 4 x 4 patch
        const int pos = i >> 10, ch = (i >> 4) & 63, e = i & 15;
        const int ky = (e >> 2) - (pos >> 1), kx = (e & 3) - (pos & 1);
        sW[i] = f2bf((ky >= 0 && ky < 3 && kx >= 0 && kx < 3) ? w[ch * 9 + ky * 3 + kx] : 0.f);
    }
    for (int i = t; i < kC1Part; i += 256) sRed[i] = 0.f;
    __syncthreads();

    f32x4 accw[4][4];                                           // dW'[pos][channel tile]: row = channel 4q + reg, col = patch element r
    _Pragma("unroll") for (int pos = 0; pos < 4; ++pos)
        _Pragma("unroll") for (int ct = 0; ct < 4; ++ct) accw[pos][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dbs[4] = {0.f, 0.f, 0.f, 0.f}, bch[4];
    _Pragma("unroll") for (int ct = 0; ct < 4; ++ct) bch[ct] = bias[16 * ct + r];
    uint16_t* mx = sX[wave];
    uint16_t* md = sD[wave];

    for (int seg = int(blockIdx.x) * 4 + wave; seg < n_seg; seg += int(gridDim.x) * 4) {
        const int n = seg / 48, py = seg % 48;
        // stage: input rows 2py-1 .. 2py+2, columns -1 .. 64 (f32 -> bf16, zeros outside the image; unconditional clamped loads) and
        // the row's 32 x 64 gradient tile (4 KiB contiguous). Three workgroups per CU hide the round trip; fetching a row ahead into
        // registers cost the third wave per SIMD and measured slower (0.75 -> 0.94 ms).
        _Pragma("unroll") for (int k = 0; k < 5; ++k) {
            const int i = lane + 64 * k, row = i / 66, col = i % 66;
            const int gy = 2 * py - 1 + row, gx = col - 1;
            const int cy = gy < 0 ? 0 : (gy > 95 ? 95 : gy), cx = gx < 0 ? 0 : (gx > 63 ? 63 : gx);
            const float v = x[(size_t(n) * 96 + cy) * 64 + cx];
            if (i < 4 * 66) mx[row * XP + col] = f2bf((gy >= 0 && gy < 96 && gx >= 0 && gx < 64) ? v : 0.f);
        }
        const u32x4* gsrc = reinterpret_cast<const u32x4*>(d_pooled + size_t(seg) * 32 * 64);
        _Pragma("unroll") for (int k = 0; k < 4; ++k) reinterpret_cast<u32x4*>(md)[lane + 64 * k] = gsrc[lane + 64 * k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the staging buffers belong to this wave; its LDS
        __builtin_amdgcn_wave_barrier();                            // operations execute in order
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // A fragments of the recompute: patch elements 8q .. 8q+7 (patch rows 2q, 2q+1) of pixel 16 pt + r; k >= 16 is padding
        bf16x8 pa[2];
        _Pragma("unroll") for (int pt = 0; pt < 2; ++pt) {
            const int px = 16 * pt + r;
            const uint32_t* r0 = reinterpret_cast<const uint32_t*>(mx + (2 * (q & 1)) * XP + 2 * px);
            const uint32_t* r1 = reinterpret_cast<const uint32_t*>(mx + (2 * (q & 1) + 1) * XP + 2 * px);
            const u32x4 ur{r0[0], r0[1], r1[0], r1[1]};
            pa[pt] = __builtin_bit_cast(bf16x8, q < 2 ? ur : u32x4{0u, 0u, 0u, 0u});
        }
        // B fragment of the gradient GEMM: patch element r of the 8 pixels {4q .. 4q+3, 16+4q .. 16+4q+3}
        bf16x8 pb;
        {
            const uint16_t* base = mx + (r >> 2) * XP + (r & 3);
            uint16_t e[8];
            _Pragma("unroll") for (int j = 0; j < 8; ++j) e[j] = base[2 * ((j < 4 ? 0 : 16) + 4 * q + (j & 3))];
            pb = __builtin_bit_cast(bf16x8, u32x4{uint32_t(e[0]) | (uint32_t(e[1]) << 16), uint32_t(e[2]) | (uint32_t(e[3]) << 16),
                                                  uint32_t(e[4]) | (uint32_t(e[5]) << 16), uint32_t(e[6]) | (uint32_t(e[7]) << 16)});
        }
        _Pragma("unroll") for (int ct = 0; ct < 4; ++ct) {
            __builtin_amdgcn_sched_barrier(0);                      // one channel tile at a time: hoisting the next tile's operand reads
                                                                    // up here costs ~90 registers (they went to AGPRs and back)
            // (1) pre^T[pixel][channel] for the four window positions; D: col = channel 16 ct + r, row = pixel 4q + reg
            f32x4 pre[4][2];
            _Pragma("unroll") for (int pos = 0; pos < 4; ++pos) {
                // K slots 16 .. 31 are padding: lanes q >= 2 read a valid address and select zero (no branch around the read)
                const u32x4 bwr = *reinterpret_cast<const u32x4*>(sW + ((pos * 64 + 16 * ct + r) * 16 + 8 * (q & 1)));
                const u32x4 bw = q < 2 ? bwr : u32x4{0u, 0u, 0u, 0u};
                _Pragma("unroll") for (int pt = 0; pt < 2; ++pt)
                    pre[pos][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[pt], __builtin_bit_cast(bf16x8, bw),
                                                                           f32x4{bch[ct], bch[ct], bch[ct], bch[ct]}, 0, 0, 0);
            }
            // (2) this lane's channel: first maximum, ReLU mask, routed gradient for its 2 x 4 pixels. Lane-mask logic (scalar ALU) picks
            // the position; the bf16 gradient halves are masked in place (dY is bf16 already: the masked copy is exact)
            uint32_t gp[4][4];                                   // [pos][pixel pair] packed bf16 pairs = the A fragment's dwords
            _Pragma("unroll") for (int pt = 0; pt < 2; ++pt) {
                const s16x4 dv = tr_read(reinterpret_cast<const char*>(md + ((16 * pt + 4 * q + q4) * 64 + 16 * ct + 4 * p4)));
                const u32x2 dw = __builtin_bit_cast(u32x2, dv);      // pixels (0, 1) and (2, 3) of this lane's channel
                _Pragma("unroll") for (int pr = 0; pr < 2; ++pr) {
                    uint32_t sel[4][2];                          // [pos][half]: the half's bits if the first maximum is at pos
                    _Pragma("unroll") for (int hf = 0; hf < 2; ++hf) {
                        const int e = 2 * pr + hf;
                        const float p0 = pre[0][pt][e], p1 = pre[1][pt][e], p2 = pre[2][pt][e], p3 = pre[3][pt][e];
                        const bool m1 = p1 > p0;
                        const float b1 = m1 ? p1 : p0;
                        const bool m2 = p2 > b1;
                        const float b2 = m2 ? p2 : b1;
                        const bool m3 = p3 > b2;
                        const float b3 = m3 ? p3 : b2;
                        const bool on = b3 > 0.f;
                        const uint32_t bits = hf ? (dw[pr] & 0xffff0000u) : (dw[pr] & 0xffffu);
                        const uint32_t gbits = on ? bits : 0u;
                        dbs[ct] += __builtin_bit_cast(float, hf ? gbits : (gbits << 16));
                        sel[3][hf] = m3 ? gbits : 0u;
                        sel[2][hf] = (m2 && !m3) ? gbits : 0u;
                        sel[1][hf] = (m1 && !m2 && !m3) ? gbits : 0u;
                        sel[0][hf] = (!m1 && !m2 && !m3) ? gbits : 0u;
                    }
                    _Pragma("unroll") for (int pos = 0; pos < 4; ++pos) gp[pos][2 * pt + pr] = sel[pos][0] | sel[pos][1];
                }
            }
            // (3) dW'[pos][channel][element] += G_pos . patch^T
            _Pragma("unroll") for (int pos = 0; pos < 4; ++pos)
                accw[pos][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8, u32x4{gp[pos][0], gp[pos][1], gp[pos][2], gp[pos][3]}), pb, accw[pos][ct], 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the reads above precede the next row's staging writes
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // db: the 4 lane groups hold different pixels of the same channel
    _Pragma("unroll") for (int ct = 0; ct < 4; ++ct) {
        dbs[ct] += __shfl_xor(dbs[ct], 16);
        dbs[ct] += __shfl_xor(dbs[ct], 32);
    }
    // the workgroup's four waves, added in order
    for (int wv = 0; wv < 4; ++wv) {
        if (wave == wv) {
            _Pragma("unroll") for (int pos = 0; pos < 4; ++pos)
                _Pragma("unroll") for (int ct = 0; ct < 4; ++ct)
                    _Pragma("unroll") for (int e = 0; e < 4; ++e)
                        sRed[(pos * 64 + 16 * ct + 4 * q + e) * 16 + r] += accw[pos][ct][e];
            if (q == 0)
                _Pragma("unroll") for (int ct = 0; ct < 4; ++ct) sRed[4096 + 16 * ct + r] += dbs[ct];
        }
        __syncthreads();
    }
    for (int i = t; i < kC1Part; i += 256) partial[size_t(blockIdx.x) * kC1Part + i] = sRed[i];
}

// dW[c][ky][kx] = sum over workgroups and positions of dW'[pos][c][4 (dy + ky) + dx + kx]; db[c]: one workgroup per output
__global__ __launch_bounds__(256) void conv1_bwd_mfma_finish_kernel(const float* __restrict__ partial, int blocks, float* __restrict__ dw,
                                                                    float* __restrict__ db) {
    __shared__ double part[256];
    const int i = blockIdx.x;                                      // 0 .. 639: channel * 10 + (tap | bias)
    const int c = i / 10, k = i % 10;
    double s = 0.0;
    for (int b = threadIdx.x; b < blocks; b += 256) {
        const float* p = partial + size_t(b) * kC1Part;
        if (k == 9) {
            s += p[4096 + c];
        } else {
            const int ky = k / 3, kx = k % 3;
            _Pragma("unroll") for (int pos = 0; pos < 4; ++pos) s += p[(pos * 64 + c) * 16 + 4 * ((pos >> 1) + ky) + (pos & 1) + kx];
        }
    }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int wv = 128; wv > 0; wv >>= 1) {
        if (int(threadIdx.x) < wv) part[threadIdx.x] += part[threadIdx.x + wv];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (k < 9) dw[c * 9 + k] = float(part[0]);
        else db[c] = float(part[0]);
    }
}

// ------------------------------------------------------------------- Linear-backward helpers ---
// out[c][r] = in[r][c] for 2-byte elements; 64 x 64 tiles through LDS
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const uint16_t* __restrict__ in, int64_t ld_in, uint16_t* __restrict__ out,
                                                             int64_t ld_out, int64_t rows, int64_t cols) {
    __shared__ uint16_t tile[64][66];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;          // 64 x 4
    const int64_t c0 = int64_t(blockIdx.x) * 64, r0 = int64_t(blockIdx.y) * 64;
    _Pragma("unroll") for (int j = 0; j < 64; j += 4) {
        const int64_t r = r0 + ty + j, c = c0 + tx;
        tile[ty + j][tx] = (r < rows && c < cols) ? in[r * ld_in + c] : uint16_t(0);
    }
    __syncthreads();
    _Pragma("unroll") for (int j = 0; j < 64; j += 4) {
        const int64_t c = c0 + ty + j, r = r0 + tx;
        if (c < cols && r < ld_out) out[c * ld_out + r] = r < rows ? tile[tx][ty + j] : uint16_t(0);     // zero the row padding too
    }
}

__global__ __launch_bounds__(256) void colsum_partial_bf16_kernel(const bf16_t* __restrict__ x, int64_t ldx, int64_t rows, int cols,
                                                                  double* __restrict__ partial) {
    __shared__ double part[4][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int64_t per = (rows + gridDim.y - 1) / gridDim.y;
    const int64_t r0 = int64_t(blockIdx.y) * per, r1 = r0 + per < rows ? r0 + per : rows;
    double s = 0.0;
    if (c < cols)
        for (int64_t r = r0 + g; r < r1; r += 4) s += double(bf2f(x[r * ldx + c].bits));
    part[g][lane] = s;
    __syncthreads();
    if (g == 0 && c < cols) partial[int64_t(blockIdx.y) * cols + c] = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
}

__global__ void colsum_finish_bf16_kernel(const double* __restrict__ partial, int chunks, int cols, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    double s = 0.0;
    for (int k = 0; k < chunks; ++k) s += partial[int64_t(k) * cols + c];
    out[c] = float(s);
}

// (Cout, Cin, 3, 3) f32 -> (Cin, 9, Cout) bf16 with the taps flipped: weights of the dgrad convolution
__global__ void repack_dgrad_bf16_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int cout, int cin) {
    const int64_t total = int64_t(cout) * 9 * cin;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += int64_t(gridDim.x) * blockDim.x) {
        const int co = int(i % cout);
        const int tap = int((i / cout) % 9);
        const int ci = int(i / (int64_t(cout) * 9));
        out[i].bits = f2bf(w[(int64_t(co) * cin + ci) * 9 + (8 - tap)]);
    }
}

constexpr int kBiasGrid8 = 1024;

}  // namespace

extern "C" int mla_maxpool2x2_bf16(const void* a, void* out, int64_t n, int H, int W, int C, mla_stream_t stream) {
    MLA_REQUIRE(a && out && n >= 0 && H % 2 == 0 && W % 2 == 0 && C % 8 == 0, MLA_E_ARG, "bad maxpool arguments");
    MLA_REQUIRE(mla::aligned(a, 16) && mla::aligned(out, 16), MLA_E_ARG, "maxpool buffers must be 16-byte aligned");
    const int64_t total = n * (H / 2) * (W / 2) * (C / 8);
    if (total == 0) return MLA_OK;
    const unsigned grid = unsigned((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(maxpool_bf16_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const bf16_t*>(a),
                       static_cast<bf16_t*>(out), total, H, W, C);
    MLA_LAUNCH_OK("maxpool_bf16");
    return MLA_OK;
}

extern "C" int64_t mla_relu_pool_bwd_bf16_workspace_bytes(void) { return int64_t(kBiasGrid8) * 256 * 8 * 8; }

extern "C" int mla_relu_pool_bwd_bf16(const void* a, int a_dtype, const void* d_out, int d_dtype, void* dz, int64_t n, int H, int W,
                                      int C, int pool, void* workspace, float* db, mla_stream_t stream) {
    MLA_REQUIRE(a && d_out && dz && n > 0, MLA_E_ARG, "bad relu_pool_bwd arguments");
    MLA_REQUIRE((a_dtype == MLA_BF16 && d_dtype == MLA_BF16) || (a_dtype == MLA_F32 && d_dtype == MLA_F32), MLA_E_DTYPE,
                "relu_pool_bwd_bf16: a / d_out dtypes %d / %d (both bf16, or both f32)", a_dtype, d_dtype);
    MLA_REQUIRE(C > 0 && C % 8 == 0 && (kBiasGrid8 * 256) % (C / 8) == 0, MLA_E_SHAPE, "channel count %d", C);
    MLA_REQUIRE(!pool || (H % 2 == 0 && W % 2 == 0), MLA_E_SHAPE, "pooling needs even H, W");
    MLA_REQUIRE(!db || workspace, MLA_E_ARG, "the bias gradient needs the workspace");
    MLA_REQUIRE(mla::aligned(a, 16) && mla::aligned(d_out, 16) && mla::aligned(dz, 16), MLA_E_ARG, "buffers must be 16-byte aligned");
    const int64_t total8 = (pool ? n * (H / 2) * (W / 2) * C : n * H * W * C) / 8;
    hipStream_t s = static_cast<hipStream_t>(stream);
    double* slots = db ? static_cast<double*>(workspace) : nullptr;
    if (a_dtype == MLA_BF16)
        hipLaunchKernelGGL((relu_pool_bwd_bf16_kernel<bf16_t, bf16_t>), dim3(kBiasGrid8), dim3(256), 0, s, static_cast<const bf16_t*>(a),
                           static_cast<const bf16_t*>(d_out), static_cast<bf16_t*>(dz), total8, H, W, C, pool, slots);
    else
        hipLaunchKernelGGL((relu_pool_bwd_bf16_kernel<float, float>), dim3(kBiasGrid8), dim3(256), 0, s, static_cast<const float*>(a),
                           static_cast<const float*>(d_out), static_cast<bf16_t*>(dz), total8, H, W, C, pool, slots);
    MLA_LAUNCH_OK("relu_pool_bwd_bf16");
    if (db) {
        hipLaunchKernelGGL(bias_slots_finish8_kernel, dim3(unsigned(C)), dim3(256), 0, s, slots, int64_t(kBiasGrid8) * 256, C, db);
        MLA_LAUNCH_OK("bias_slots_finish8");
    }
    return MLA_OK;
}

extern "C" int mla_pool_bwd_codes_bf16(const void* codes, const void* d_pooled, void* dz, int64_t n, int H, int W, int C, void* workspace,
                                       float* db, mla_stream_t stream) {
    MLA_REQUIRE(codes && d_pooled && dz && n > 0, MLA_E_ARG, "bad pool_bwd_codes arguments");
    MLA_REQUIRE(C > 0 && C % 8 == 0 && (kBiasGrid8 * 256) % (C / 8) == 0, MLA_E_SHAPE, "channel count %d", C);
    MLA_REQUIRE(H % 2 == 0 && W % 2 == 0, MLA_E_SHAPE, "pooling needs even H, W");
    MLA_REQUIRE(!db || workspace, MLA_E_ARG, "the bias gradient needs the workspace");
    MLA_REQUIRE(mla::aligned(codes, 8) && mla::aligned(d_pooled, 16) && mla::aligned(dz, 16), MLA_E_ARG, "buffers must be 16-byte aligned (codes: 8)");
    const int64_t total8 = n * (H / 2) * (W / 2) * C / 8;
    hipStream_t s = static_cast<hipStream_t>(stream);
    double* slots = db ? static_cast<double*>(workspace) : nullptr;
    hipLaunchKernelGGL(pool_bwd_codes_bf16_kernel, dim3(kBiasGrid8), dim3(256), 0, s, static_cast<const uint8_t*>(codes),
                       static_cast<const bf16_t*>(d_pooled), static_cast<bf16_t*>(dz), total8, H, W, C, slots);
    MLA_LAUNCH_OK("pool_bwd_codes_bf16");
    if (db) {
        hipLaunchKernelGGL(bias_slots_finish8_kernel, dim3(unsigned(C)), dim3(256), 0, s, slots, int64_t(kBiasGrid8) * 256, C, db);
        MLA_LAUNCH_OK("bias_slots_finish8");
    }
    return MLA_OK;
}

extern "C" int mla_conv_wgrad_bf16(const void* dz, const void* a_in, int64_t n, int H, int W, int cin, int cout, float* workspace,
                                   int64_t workspace_floats, float* dw_oihw, mla_stream_t stream) {
    MLA_REQUIRE(dz && a_in && workspace && dw_oihw && n > 0, MLA_E_ARG, "bad wgrad arguments");
    MLA_REQUIRE(mla::aligned(dz, 16) && mla::aligned(a_in, 16), MLA_E_ARG, "wgrad operands must be 16-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bf16_t* z = static_cast<const bf16_t*>(dz);
    const bf16_t* a = static_cast<const bf16_t*>(a_in);
#define MLA_WGRAD_CASE(CI, CO, HH, WW) \
    if (cin == CI && cout == CO && H == HH && W == WW) return launch_wgrad_bf16<CI, CO, HH, WW>(z, a, n, workspace, workspace_floats, dw_oihw, s);
    MLA_WGRAD_CASE(64, 128, 48, 32)
    MLA_WGRAD_CASE(128, 256, 24, 16)
    MLA_WGRAD_CASE(256, 256, 24, 16)
    MLA_WGRAD_CASE(256, 512, 12, 8)
    MLA_WGRAD_CASE(512, 512, 12, 8)
#undef MLA_WGRAD_CASE
    return mla::fail(MLA_E_SHAPE, "wgrad %dx%d %d->%d is not compiled", H, W, cin, cout);
}

extern "C" int64_t mla_conv1_bwd_workspace_floats(void) {
    const int64_t a = int64_t(1024) * 8 * 80, b = int64_t(kC1MaxWg) * kC1Part;       // f32 form (cnn_train.hip) / bf16 MFMA form
    return a > b ? a : b;
}

extern "C" int mla_conv1_bwd_bf16(const float* x, const float* w, const float* bias, const void* d_pooled, int64_t n, float* workspace,
                                  float* dw, float* db, mla_stream_t stream) {
    MLA_REQUIRE(x && w && bias && d_pooled && workspace && dw && db && n > 0, MLA_E_ARG, "bad conv1_bwd arguments");
    const int64_t n_pix = n * 48 * 32;
    hipStream_t s = static_cast<hipStream_t>(stream);
#if MLA_CONV1_BWD_MFMA
    MLA_REQUIRE(n * 48 <= 0x7fffffff, MLA_E_SHAPE, "too many rows for one launch");
    const int n_seg = int(n * 48);
    const int wgs = (n_seg + 3) / 4 < kC1MaxWg ? (n_seg + 3) / 4 : kC1MaxWg;      // workspace: kC1MaxWg * kC1Part floats
    hipLaunchKernelGGL(conv1_bwd_mfma_kernel, dim3(wgs), dim3(256), 0, s, x, w, bias, static_cast<const bf16_t*>(d_pooled), n_seg, workspace);
    MLA_LAUNCH_OK("conv1_bwd_mfma");
    hipLaunchKernelGGL(conv1_bwd_mfma_finish_kernel, dim3(640), dim3(256), 0, s, workspace, wgs, dw, db);
    MLA_LAUNCH_OK("conv1_bwd_mfma_finish");
    return MLA_OK;
#else
    const int blocks = int((n_pix + 255) / 256 < 1024 ? (n_pix + 255) / 256 : 1024);
    hipLaunchKernelGGL(conv1_bwd_bf16_kernel, dim3(blocks, 8), dim3(256), 0, s, x, w, bias, static_cast<const bf16_t*>(d_pooled), n_pix, workspace);
    MLA_LAUNCH_OK("conv1_bwd_bf16");
    hipLaunchKernelGGL(conv1_bwd_finish_bf16_kernel, dim3(640), dim3(256), 0, s, workspace, blocks, dw, db);
    MLA_LAUNCH_OK("conv1_bwd_finish_bf16");
    return MLA_OK;
#endif
}

extern "C" int mla_transpose_bf16(const void* in, int64_t ld_in, void* out, int64_t ld_out, int64_t rows, int64_t cols,
                                  mla_stream_t stream) {
    MLA_REQUIRE(in && out && rows > 0 && cols > 0 && ld_in >= cols && ld_out >= rows, MLA_E_ARG, "bad transpose arguments");
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3(unsigned((cols + 63) / 64), unsigned((ld_out + 63) / 64)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), static_cast<const uint16_t*>(in), ld_in, static_cast<uint16_t*>(out), ld_out, rows, cols);
    MLA_LAUNCH_OK("transpose_bf16");
    return MLA_OK;
}

// workspace: 64 * cols doubles
extern "C" int mla_col_sum_bf16(const void* x, int64_t ldx, int64_t rows, int64_t cols, void* workspace, float* out,
                                mla_stream_t stream) {
    MLA_REQUIRE(x && workspace && out && rows > 0 && cols > 0 && ldx >= cols, MLA_E_ARG, "bad col_sum arguments");
    const int chunks = int(rows / 256 < 1 ? 1 : (rows / 256 > 64 ? 64 : rows / 256));
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(colsum_partial_bf16_kernel, dim3(unsigned((cols + 63) / 64), unsigned(chunks)), dim3(256), 0, s,
                       static_cast<const bf16_t*>(x), ldx, rows, int(cols), static_cast<double*>(workspace));
    MLA_LAUNCH_OK("colsum partial bf16");
    hipLaunchKernelGGL(colsum_finish_bf16_kernel, dim3(unsigned((cols + 255) / 256)), dim3(256), 0, s,
                       static_cast<const double*>(workspace), chunks, int(cols), out);
    MLA_LAUNCH_OK("colsum finish bf16");
    return MLA_OK;
}

extern "C" int mla_conv_repack_dgrad_bf16(const float* w_oihw, int64_t cout, int64_t cin, void* out, mla_stream_t stream) {
    MLA_REQUIRE(w_oihw && out && cout > 0 && cin > 0, MLA_E_ARG, "bad repack arguments");
    const int64_t total = cout * 9 * cin;
    const unsigned grid = unsigned((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(repack_dgrad_bf16_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), w_oihw,
                       static_cast<bf16_t*>(out), int(cout), int(cin));
    MLA_LAUNCH_OK("repack_dgrad_bf16_kernel");
    return MLA_OK;
}
