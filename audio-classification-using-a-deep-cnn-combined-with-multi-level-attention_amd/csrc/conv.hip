// conv.hip -- VGGish feature stack on gfx950 (reference: vggish.py:108-118 make_layers(),
// applied at vggish.py:22): six 3x3/pad-1 convolutions + ReLU with four 2x2 max-pools, NHWC.
//
//   conv1 (Cin = 1)      dedicated VALU kernel, fused bias + ReLU + 2x2 pool: 0.4 % of the FLOPs
//                        but the largest activation, so it is written for the memory pipe
//                        (one pooled pixel x 64 channels per lane, 128 B contiguous stores).
//   conv2 .. conv4_2     implicit GEMM on the matrix cores: M = pixels, N = Cout, K = 9 Cin.
//                        bf16 (v_mfma_f32_16x16x32_bf16) or exact-f32 (v_mfma_f32_16x16x4_f32)
//                        from ONE source; fused bias + ReLU (+ 2x2 max-pool) epilogue, so the
//                        un-pooled activation never goes to HBM and the NHWC output makes the
//                        reference's NCHW->NHWC flatten (vggish.py:26-29) a no-op.
//
// Roofline: MFMA. Algorithmic work 2 * H*W * Cout * 9*Cin FLOP per frame and layer
// (SURVEY.md section 8a, row a9): 226.5 / 226.5 / 453.0 / 226.5 / 453.0 MFLOP for conv2..conv4_2.
//
// Workgroup = 512 threads = 8 waves; each wave owns 6 x NS 16x16 accumulator tiles (96 pixels x 16 NS channels).
// bf16: waves 4 (M) x 2 (N), "tall" output tile 384 pixels x 128 channels; f32: 2 (M) x 4 (N), 192 pixels x (64 * NS)
// channels (struct Cfg, WM). The input patch (tile rows + halo,
// one 128-byte channel chunk per pixel) is parked in LDS ONCE per channel chunk and serves all
// nine taps as shifted reads -- no im2col copy; weights stream through a double-buffered LDS
// tile per (tap, chunk) with the next slice in flight (registers) behind the MFMAs. M-subtiles
// are laid out so that 2x2 pooling is lane-local in the accumulator registers:
//   W >= 16 : an m-subtile = 16 consecutive x of one image row; rows y / y+1 are the wave's
//             subtiles i / i+1, x pairs are accumulator registers (0,1) / (2,3);
//   W == 8  : an m-subtile = 8 x of image a + 8 x of image a+1 (same row).
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "mma_core.h"

namespace {

using namespace mma;

constexpr int kThreads = 512, kWaves = 8, kMS = 6;

// Diagnostic build only (-DMLA_CONV_STAMPS=1, scripts/build_variant.py): s_memtime stamps of ONE tap (tap 4 of a workgroup's
// second channel chunk) of waves 0 and 4 of the first 8 workgroups, written to a buffer nothing else reads; read back through
// mla_debug_conv_stamps(). The shipped kernel executes no stamp.
#ifndef MLA_CONV_STAMPS
#define MLA_CONV_STAMPS 0
#endif
#ifndef MLA_STAMP_TAP
#define MLA_STAMP_TAP 4
#define MLA_STAMP_CHUNK 5
#endif
#if MLA_CONV_STAMPS
__device__ unsigned long long g_conv_stamps[8][2][8];
#define MLA_STAMP(k) do { if (stamp_on) stamps[k] = __builtin_amdgcn_s_memtime(); } while (0)
#if MLA_CONV_STAMPS == 2
#undef MLA_STAMP
#define MLA_STAMP(k) do { } while (0)
#define MLA_STAMP2(k) do { if (stamp_on) stamps[k] = __builtin_amdgcn_s_memtime(); } while (0)
#endif
#else
#define MLA_STAMP(k) do { } while (0)
#endif

#ifndef MLA_CONV_PRIO
#define MLA_CONV_PRIO 3             // 3 (shipped): burst priorities 3 / 2 / 1 / 0 so that no MFMA burst is preempted; 1: waves 4-7 static prio 1; 0: none
#endif
#ifndef MLA_CONV_DGRAD2
#define MLA_CONV_DGRAD2 2           // dgrad conv2 (128 -> 64 @48x32, bf16) tile: 0 = 192 px x 64 ch (2 x 4 waves, NS 1), two workgroups per CU;
                                    // 1 = 384 px x 64 ch (4 x 2 waves, NS 2), two workgroups per CU; 2 = the same tile as ONE persistent workgroup
                                    // per CU (LDS-DMA patches, staggered wave pairs). Same device, 5 120 images: 1.38 / 1.69 / 1.15 ms -> 2 (bit-identical)
#endif
#ifndef MLA_CONV_NARROW_PERSIST
#define MLA_CONV_NARROW_PERSIST 0   // 1: conv2's bf16 inference configuration as one persistent workgroup per CU (A/B builds)
#endif
#ifndef MLA_CONV1_RAW_BARRIER
#define MLA_CONV1_RAW_BARRIER 1
#endif
#ifndef MLA_CONV1_WAVES
#define MLA_CONV1_WAVES 3
#endif
#ifndef MLA_CONV1_ROWS
#define MLA_CONV1_ROWS 16            // input rows per workgroup tile of the patch-GEMM conv1 (must divide 96; measured 8: 0.87, 16: 0.43, 24 / 32 / 48: 0.56-0.73 ms)
#endif
#ifndef MLA_CONV1_GATHER
#define MLA_CONV1_GATHER 0          // 1: bf16 conv1 through the round-1 gather kernel (A/B builds)
#endif
#ifndef MLA_CONV_TALL
#define MLA_CONV_TALL 2              // bf16 layers on 384 x 128 tiles: 1 = conv2..conv4 (W >= 16), 2 = conv5 / conv6 (W = 8) too; 0: A/B builds on the 192-pixel tiles
#endif
#ifndef MLA_CONV_PATCH_SPREAD
#define MLA_CONV_PATCH_SPREAD 1
#endif
#ifndef MLA_CONV_TALL_SPLIT
#define MLA_CONV_TALL_SPLIT 1
#endif
#ifndef MLA_CONV_NT_STORE
#define MLA_CONV_NT_STORE 1              // streaming outputs (conv epilogues, conv1) as non-temporal stores: conv1 0.43 -> 0.38 ms, conv3 -4 % in the pipeline
#endif
#ifndef MLA_CONV_DMA_DIV
#define MLA_CONV_DMA_DIV 1
#endif
#ifndef MLA_CONV_DMA_LATE
#define MLA_CONV_DMA_LATE 1
#endif
#ifndef MLA_CONV_STAGGER
#define MLA_CONV_STAGGER 1          // 0: A/B builds without the half-tap stagger of waves 4-7 (scripts/build_variant.py)
#endif

// SPLIT ("bf16x3"): every f32 value x travels as two bf16, hi = bf16(x) and lo = bf16(x - hi) (x = hi + lo to 2^-18
// relative). Activations hold [hi(C) | lo(C)] per pixel, repacked weights [hi | lo | hi] per 64-channel chunk and tap, and
// the K loop runs the three products a_hi w_hi + a_hi w_lo + a_lo w_hi (bf16 products are exact in the f32 accumulator;
// the dropped a_lo w_lo is 2^-18 relative): f32-grade results at a third of the bf16 MFMA rate.
// WM: waves along M. 2 (x 4 along N): tile 192 pixels x 64 NS channels. 4 (x 2 along N, "tall"): 384 pixels x 32 NS channels -- the same
// 6 x NS accumulator tiles per wave, but a weight slice (the operand that is re-fetched for every tap) is half as large per FLOP:
// L2 -> LDS traffic per tap 22.7 instead of 35.2 KB (conv4), which this power-limited kernel returns as clock (section 3.3 of DESIGN.md).
template <typename T, int CIN_, int COUT_, int H_, int W_, bool POOL_, int NS_, bool ACT_ = true, bool SPLIT_ = false, int WM_ = 2>
struct Cfg {
    static constexpr int WM = WM_, WN = kWaves / WM_;
    using elem = T;
    static constexpr int CIN = CIN_, COUT = COUT_, H = H_, W = W_, NS = NS_;
    static constexpr bool POOL = POOL_;
    static constexpr bool ACT = ACT_;                       // epilogue: bias + ReLU (forward) or plain store (dgrad)
    static constexpr bool SPLIT = SPLIT_;
    static constexpr int CIN_A = SPLIT ? 2 * CIN : CIN;     // channels per input pixel in memory
    static constexpr int CIN_W = SPLIT ? 3 * CIN : CIN;     // channels per tap of the repacked weights = K per tap
    static constexpr int COUT_MEM = SPLIT ? 2 * COUT : COUT;
    static_assert(!SPLIT || (sizeof(T) == 2 && ACT_), "the split mode is a bf16 forward mode");
    static constexpr int SEGW = W >= 16 ? 16 : 8;          // pixels of one image row per m-subtile
    static constexpr int SEGS = W / SEGW;                   // subtiles per tile row (2 for W = 32)
    static constexpr int IMGS = SEGW == 8 ? WM : 1;         // images per tile (W = 8: one image pair per two waves along M)
    static constexpr int TH = SEGW == 8 ? 12 : kMS * WM / SEGS;   // tile rows: 6 m-subtiles per wave along M
    // patch with halo; PW is the row PITCH. Four W = 8 images (tall tile) only fit LDS twice at pitch 9: a row's right halo pixel
    // IS the next row's left one (both are zero), one extra pixel closes the last row. The swizzle depends on the column only and
    // the images of a subtile are 14 rows = 126 pixels (even) apart, so a read class still sees 8 consecutive columns of one row
    // parity = 8 distinct bank slots (mma_core.h).
    static constexpr int PW = (W == 8 && WM == 4) ? 9 : W + 2, PH = TH + 2;
    static constexpr int BN = WN * NS * 16;
    static constexpr int KC = Elem<T>::kPerRow;             // channels per 128-byte chunk
    static constexpr int A_PIX = IMGS * PH * PW + (PW & 1); // patch pixels (one 128-byte row each)
    static constexpr int A_BYTES = (A_PIX + 7) / 8 * 8 * kRowBytes;   // padded to whole 1 KiB LDS-DMA pieces
    static constexpr int B_BYTES = BN * kRowBytes;
    static constexpr int TILES_Y = H / TH;
    static constexpr bool PERSIST = NS > 2 || SPLIT || (MLA_CONV_NARROW_PERSIST && sizeof(T) == 2 && NS == 2 && POOL_ && ACT_) ||
                                    (MLA_CONV_DGRAD2 == 2 && sizeof(T) == 2 && NS == 2 && WM_ == 4 && !ACT_ && COUT_ == 64);        // conv2 (half-width tile): two non-persistent workgroups per CU
                                                            // (its split form has a 3x longer K loop and more epilogue registers)
    static constexpr int MIN_WAVES = (PERSIST || sizeof(T) == 4) ? 2 : 4;      // waves per SIMD the register budget is held to
    static constexpr bool A_DMA = PERSIST;                  // input patches by LDS-DMA into two alternating buffers (else: one
                                                            // buffer, register-staged -- the two-workgroups-per-CU configurations)
    static constexpr int A_BUFS = A_DMA ? 2 : 1;
    // the next patch's DMA pieces one per wave and tap instead of all in tap 2: measured per shape (same-device A/B) +3.2 % on the
    // tall conv4, -1 ... -2 % on conv3 / conv5 (register allocation shifts), neutral elsewhere
    static constexpr bool PATCH_SPREAD = MLA_CONV_PATCH_SPREAD && WM_ == 4 && CIN_ >= 256;
    // Stagger (one workgroup per CU = two waves per SIMD that would otherwise run in lockstep): waves 4-7 -- the SIMD partners
    // of waves 0-3 -- run half a tap behind. Their k-step-1 fragments are READ before the barrier that ends a tap (the slice is
    // still valid there) and MULTIPLIED after it, so right after every barrier one wave of each SIMD has a full burst of MFMAs
    // ready while its partner waits for the fragments of the next tap: the matrix pipe no longer idles behind each barrier.
    // Every accumulator still receives its products in the same order: results are bit-identical.
    static constexpr bool STAGGER = PERSIST && MLA_CONV_STAGGER;
    static constexpr int LDS_BYTES = A_BUFS * A_BYTES + 2 * B_BYTES;
    static constexpr int HO = POOL ? H / 2 : H, WO = POOL ? W / 2 : W;
    static_assert(W == 32 || W == 16 || W == 8, "tile mapping covers the VGGish widths");
    static_assert(WM == 2 || WM == 4, "waves are laid out 2 x 4 or 4 x 2");
    static_assert(H % TH == 0 && CIN % KC == 0 && COUT % BN == 0, "shape must tile exactly");
    static_assert((PW * kRowBytes) % 256 == 0 || (W == 8 && (PH * PW) % 2 == 0), "row pitch / image distance must keep the bank swizzle invariant");
    static_assert(LDS_BYTES <= 160 * 1024, "tile does not fit LDS");
};

// N consecutive output channels of one pixel as ONE store (N = 2 or 4; p is N * sizeof(T) aligned)
template <typename T, int N> __device__ __forceinline__ void store_vec(T* p, const float* v);
template <> __device__ __forceinline__ void store_vec<float, 4>(float* p, const float* v) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
}
template <> __device__ __forceinline__ void store_vec<float, 2>(float* p, const float* v) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    *reinterpret_cast<f32x2*>(p) = f32x2{v[0], v[1]};
}
template <> __device__ __forceinline__ void store_vec<bf16_t, 4>(bf16_t* p, const float* v) {
#if MLA_CONV_NT_STORE
    __builtin_nontemporal_store(u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])}, reinterpret_cast<u32x2*>(p));
#else
    *reinterpret_cast<u32x2*>(p) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
#endif
}
template <> __device__ __forceinline__ void store_vec<bf16_t, 2>(bf16_t* p, const float* v) {
    *reinterpret_cast<uint32_t*>(p) = pack_bf16x2(v[0], v[1]);
}

template <> __device__ __forceinline__ void store_vec<float, 1>(float* p, const float* v) { *p = v[0]; }
template <> __device__ __forceinline__ void store_vec<bf16_t, 1>(bf16_t* p, const float* v) { p->bits = f2bf(v[0]); }

// the lane's NS channels of one output pixel: one vector store, or (SPLIT) the hi plane and the lo plane COUT apart
template <typename C>
__device__ __forceinline__ void store_px(typename C::elem* p, const float* v) {
    using T = typename C::elem;
    if constexpr (C::SPLIT) {
        float hi[C::NS], lo[C::NS];
        _Pragma("unroll") for (int j = 0; j < C::NS; ++j) {
            hi[j] = bf2f(f2bf(v[j]));
            lo[j] = v[j] - hi[j];
        }
        store_vec<T, C::NS>(p, hi);                 // exact: hi is a bf16 value
        store_vec<T, C::NS>(p + C::COUT, lo);
    } else {
        store_vec<T, C::NS>(p, v);
    }
}

template <typename C>
__device__ __forceinline__ int a_swizzle(int xh, int img) {
    // bits 1-2 only (see mma_core.h); for W = 8 the two images of a subtile occupy complementary
    // x ranges inside each read class, so no image bit is needed.
    (void)img;
    return ((xh >> 1) & 3) << 1;
}

template <typename C>
__global__ __launch_bounds__(kThreads, C::MIN_WAVES) void conv3x3_kernel(const typename C::elem* __restrict__ in,
                                                               const typename C::elem* __restrict__ wgt,
                                                               const float* __restrict__ bias,
                                                               typename C::elem* __restrict__ out, int n_img, int n_tiles,
                                                               typename C::elem* __restrict__ prepool, uint8_t* __restrict__ codes) {
    using T = typename C::elem;
    constexpr int PER = Elem<T>::kPerChunk;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;
    char* sB = smem + C::A_BUFS * C::A_BYTES;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave / C::WN, wn = wave % C::WN;
    const int r = lane & 15, q = lane >> 4;
    // Persistent workgroup: tiles blockIdx.x, blockIdx.x + gridDim.x, ... (gridDim.x is a multiple of
    // TILES_Y, so the tile row `ty` -- and with it every per-lane global offset -- is fixed for life).
    const int ty = blockIdx.x % C::TILES_Y;
    const int y_tile = ty * C::TH;
    const int n0 = blockIdx.y * C::BN;
    int tile = blockIdx.x;
    int img0 = (tile / C::TILES_Y) * C::IMGS;

    // this lane's pixel inside the tile (A-operand row r of every m-subtile of the wave)
    const int wimg = C::SEGW == 8 ? 2 * (wm >> 1) : 0;      // W = 8: waves 2k, 2k+1 along M share image pair k
    // the wave's 6 subtiles: rows l_y0 .. l_y0 + 5 of the tile, x half wxh of a 32-wide row
    const int wxh = C::SEGS == 2 ? (wm & 1) : 0, wy = C::SEGS == 2 ? (wm >> 1) : (C::SEGW == 8 ? (wm & 1) : wm);
    const int l_img = C::SEGW == 8 ? wimg + (r >> 3) : 0;
    const int l_x = C::SEGW == 8 ? (r & 7) : (wxh * 16 + r);
    const int l_y0 = kMS * wy;
    int abase[3];
    _Pragma("unroll") for (int kx = 0; kx < 3; ++kx) {
        const int xh = l_x + kx;
        abase[kx] = ((l_img * C::PH + l_y0) * C::PW + xh) * kRowBytes + 16 * (q ^ a_swizzle<C>(xh, l_img));
    }
    const int bbase = tile_off(wn * C::NS * 16 + r, q);

    // The accumulators start at the bias (a lane's four values of a subtile share its channel), so the epilogue is a
    // bare ReLU (+ pool): one VALU pass less per tile. The lane's NS accumulators of a pixel are NS consecutive channels
    // (see b_dma).
    const int nb = n0 + (wn * 16 + r) * C::NS;
    float bv[C::NS];
    _Pragma("unroll") for (int j = 0; j < C::NS; ++j) bv[j] = C::ACT ? bias[nb + j] : 0.f;
    f32x4 acc[kMS][C::NS];
    _Pragma("unroll") for (int i = 0; i < kMS; ++i)
        _Pragma("unroll") for (int j = 0; j < C::NS; ++j) acc[i][j] = f32x4{bv[j], bv[j], bv[j], bv[j]};

    constexpr int A_PIECES = C::IMGS * C::PH * C::PW * 8;
    constexpr int A_PASSES = (A_PIECES + kThreads - 1) / kThreads;
    constexpr int CHUNKS = C::CIN_W / C::KC;                // K chunks per tap (three passes over the input planes when SPLIT)
    constexpr int PLANE = C::CIN / C::KC;                    // chunks per input plane
    // input channel offset that pairs with weight chunk c. SPLIT: the weights hold [w_hi | w_lo | w_hi] per 64-channel
    // chunk k, paired with [a_hi_k | a_hi_k | a_lo_k]: the hi patch staged for pass 0 is reused by pass 1
    auto a_chan = [](int c) { return (!C::SPLIT ? c : (c % 3 < 2 ? c / 3 : PLANE + c / 3)) * C::KC; };

    // Global operands through buffer descriptors (wave-uniform base, 32-bit per-lane offsets fixed
    // for the whole kernel, scalar offset per tap / chunk). Out-of-image halo pixels and images past
    // the batch get an out-of-range offset: the hardware range check returns zeros, no branches.
    constexpr uint32_t ESZ = sizeof(T);
    auto patch_rsrc = [&](int first_img) {     // descriptor over the tile's images (fewer at the batch tail)
        const int imgs_here = (n_img - first_img) < C::IMGS ? (n_img - first_img) : C::IMGS;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(in) + size_t(first_img) * C::H * C::W * C::CIN_A, 0,
                                                 uint32_t(imgs_here) * C::H * C::W * C::CIN_A * ESZ, 0x00020000);
    };
    __amdgpu_buffer_rsrc_t a_rsrc = patch_rsrc(img0);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(wgt) + size_t(n0) * 9 * C::CIN_W, 0, uint32_t(C::BN) * 9 * C::CIN_W * ESZ, 0x00020000);
    // Per-lane offsets are recomputed where they are used (once per chunk for the patch, ~2 VALU per
    // weight piece) instead of being held in VGPRs across the MFMA loop: registers are the scarce
    // resource here (96 accumulators + 40 fragment + 36 staging registers per lane).
    u32x4 apre[A_PASSES];
    auto a_load = [&](int c0) {            // input patch of one channel chunk -> registers
        _Pragma("unroll") for (int p = 0; p < A_PASSES; ++p) {
            const int piece = t + kThreads * p;
            const int ch = piece & 7, pix = piece >> 3;
            const int xh = pix % C::PW, rest = pix / C::PW;
            const int yh = rest % C::PH, im = rest / C::PH;
            const int gy = y_tile + yh - 1, gx = xh - 1;
            const bool ok = piece < A_PIECES && gy >= 0 && gy < C::H && gx >= 0 && gx < C::W;
            const int voff = ok ? int((((im * C::H + gy) * C::W + gx) * C::CIN_A + ch * PER) * ESZ) : int(0x7fffff00);
            apre[p] = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, voff, int(c0 * ESZ), 0);
        }
    };
    auto a_write = [&]() {
        _Pragma("unroll") for (int p = 0; p < A_PASSES; ++p) {
            const int piece = t + kThreads * p;
            const int ch = piece & 7, pix = piece >> 3;
            const int xh = pix % C::PW, rest = pix / C::PW;
            const int yh = rest % C::PH, im = rest / C::PH;
            if (piece < A_PIECES)
                lds_write16(sA, ((im * C::PH + yh) * C::PW + xh) * kRowBytes + 16 * (ch ^ a_swizzle<C>(xh, im)), apre[p]);
        }
    };
    // One (tap, chunk) weight slice (BN rows x 128 B) goes global -> LDS by LDS-DMA (buffer_load ... lds): no staging
    // registers, no ds_write pass. A wave-instruction fills 1 KiB = 8 consecutive tile rows in lane order, so the
    // tile's XOR swizzle is applied on the SOURCE side: the lane at (row, slot c') fetches chunk c' ^ swz(row).
    // LDS row j*16 + r of a wave's NS*16-row block holds output channel r*NS + j of that block: the NS accumulators
    // of a lane are then NS consecutive channels and leave as one vector store per pixel.
    auto b_dma = [&](int c0, int tap, int buf_off) {
        _Pragma("unroll") for (int p = 0; p < C::BN / 64 / MLA_CONV_DMA_DIV; ++p) {  // DIV > 1: timing experiment only (wrong results)
            const int row = 8 * (wave + 8 * p) + (lane >> 3), slot = lane & 7;
            const int chunk = slot ^ (((row >> 1) & 3) << 1);                    // inverse of tile_off's swizzle (an involution)
            const int n = (row & ~(C::NS * 16 - 1)) + (row & 15) * C::NS + ((row >> 4) & (C::NS - 1));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, (__attribute__((address_space(3))) void*)(sB + buf_off + 8 * (wave + 8 * p) * kRowBytes),
                                                 16, int((n * 9 * C::CIN_W + chunk * PER) * ESZ), int((tap * C::CIN_W + c0) * ESZ), 0, 0);
        }
    };

    // Input patch of one channel chunk by LDS-DMA (A_DMA): eight patch pixels per wave-instruction, swizzle on the
    // source side. Out-of-image halo pixels (and images past the batch) carry an out-of-range offset: the range check
    // DROPS those lanes, so the halo rows of both patch buffers are zeroed once, below, and never written again -- they
    // are the same LDS rows for every tile of this workgroup (its tile row is fixed, the x halo always is).
    constexpr int A_INSTRS = (C::A_PIX + 7) / 8;          // 1 KiB DMA pieces per patch
    constexpr int A_ROUNDS = (A_INSTRS + 7) / 8;          // one piece per wave and round
    static_assert(A_ROUNDS <= 9, "a patch is staged within the nine taps of the previous chunk");
    auto a_dma_round = [&](int c0, int abuf, int p) {
        const int wi = wave + 8 * p;                           // wave-uniform piece index
        if (wi < A_INSTRS) {
            const int pix = 8 * wi + (lane >> 3), slot = lane & 7;
            const int xh = pix % C::PW, rest = pix / C::PW;
            const int yh = rest % C::PH, im = rest / C::PH;
            const int gy = y_tile + yh - 1, gx = xh - 1;
            const bool ok = pix < C::A_PIX && gy >= 0 && gy < C::H && gx >= 0 && gx < C::W;
            const int chunk = slot ^ a_swizzle<C>(xh, im);
            const int voff = ok ? int((((im * C::H + gy) * C::W + gx) * C::CIN_A + chunk * PER) * ESZ) : int(0x7fffff00);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (__attribute__((address_space(3))) void*)(sA + abuf * C::A_BYTES + wi * 8 * kRowBytes),
                                                     16, voff, int(c0 * ESZ), 0, 0);
        }
    };
    auto a_dma = [&](int c0, int abuf) {
        _Pragma("unroll") for (int p = 0; p < A_ROUNDS; ++p) a_dma_round(c0, abuf, p);
    };

    // prologue: first patch and first weight slice
    int abuf = 0;
    if constexpr (C::A_DMA) {
        for (int i = t; i < C::A_BUFS * C::A_BYTES / 16; i += kThreads) lds_write16(sA, 16 * i, zero16());
        __syncthreads();                           // zeros are in place before any DMA can land
        a_dma(0, 0);
        b_dma(0, 0, 0);
    } else {
        a_load(0);
        b_dma(0, 0, 0);
        a_write();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // One barrier per tap. The weight pipeline runs continuously across channel chunks AND across tiles (slice g+1 is
    // DMA'd into the other LDS buffer under the MFMAs of slice g); the next chunk's / next tile's input patch is DMA'd
    // into the other patch buffer early in the current chunk (register-staged and swapped in behind one extra barrier
    // where only one patch buffer fits), so no global round trip is exposed at a boundary.
    int par = 0;                                   // parity of the running tap counter -> current weight buffer
    const bool late = C::STAGGER && __builtin_amdgcn_readfirstlane(wave) >= 4;
#if MLA_CONV_PRIO == 1 || MLA_CONV_PRIO == 3
    if (C::PERSIST && __builtin_amdgcn_readfirstlane(wave) >= 4) __builtin_amdgcn_s_setprio(1);
#elif MLA_CONV_PRIO == 2
    if (C::PERSIST && __builtin_amdgcn_readfirstlane(wave) < 4) __builtin_amdgcn_s_setprio(1);
#endif
    // two copies of the tile loop, selected once per wave (a scalar branch): LATE = the staggered schedule of waves 4-7
#if MLA_CONV_STAMPS
    unsigned long long stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int chunk_counter = 0;
#endif
    auto run = [&](auto late_c) {
    constexpr bool LATE = decltype(late_c)::value;
    u32x4 af[kMS], bf[C::NS];                      // fragments of one k-step (late waves carry them across the barrier)
    for (;;) {
        const int next_tile = tile + int(gridDim.x);
        const bool has_next = C::PERSIST && next_tile < n_tiles;
        for (int c = 0; c < CHUNKS; ++c) {
            const int c0 = c * C::KC;
            const bool last_chunk = c + 1 == CHUNKS;
            const bool more = !last_chunk || has_next;                       // another stage follows
            const bool new_patch = more && !(C::SPLIT && !last_chunk && c % 3 == 0);   // the next stage needs other input channels
            _Pragma("unroll") for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap % 3;
                const int cur = par ? C::B_BYTES : 0;
                // next weight slice into the other buffer (its last readers passed the barrier that ended the previous tap) and, once
                // per chunk, the next patch. Issued AFTER the wave's first fragment reads (MLA_CONV_DMA_LATE): a DMA piece costs the
                // issuing wave 60-185 cycles, the data is not needed before the end of the tap, and the partner of a wave that
                // reads first gets the matrix pipe ~300 cycles earlier (in-kernel stamps: profiles/r02_conv_stamps.txt)
                auto issue_dma = [&]() {
                    if (tap < 8) b_dma(c0, tap + 1, cur ^ C::B_BYTES);
                    else if (more) b_dma(last_chunk ? 0 : c0 + C::KC, 0, cur ^ C::B_BYTES);
                    if constexpr (C::A_DMA) {
                        // next chunk's / tile's patch into the other patch buffer (idle since the previous chunk), one piece per wave
                        // and tap (MLA_CONV_PATCH_SPREAD) so that no tap carries the whole patch's DMA issue
                        if constexpr (C::PATCH_SPREAD) {
                          if (tap < A_ROUNDS && new_patch) {
                            if (tap == 0 && last_chunk) a_rsrc = patch_rsrc((next_tile / C::TILES_Y) * C::IMGS);
                            a_dma_round(last_chunk ? 0 : a_chan(c + 1), abuf ^ 1, tap);
                          }
                        } else if (tap == 2 && new_patch) {
                            if (last_chunk) a_rsrc = patch_rsrc((next_tile / C::TILES_Y) * C::IMGS);
                            a_dma(last_chunk ? 0 : a_chan(c + 1), abuf ^ 1);
                        }
                    } else {
                        if (tap == 5 && new_patch) {
                            if (last_chunk) a_rsrc = patch_rsrc((next_tile / C::TILES_Y) * C::IMGS);
                            a_load(last_chunk ? 0 : a_chan(c + 1));
                        }
                    }
                };
                auto rd = [&](int ks) {
                    _Pragma("unroll") for (int i = 0; i < kMS; ++i)
                        af[i] = lds_read16(sA + abuf * C::A_BYTES, (abase[kx] ^ (ks << 6)) + (i + ky) * C::PW * kRowBytes);
                    _Pragma("unroll") for (int j = 0; j < C::NS; ++j)
                        bf[j] = lds_read16(sB, cur + (bbase ^ (ks << 6)) + j * 16 * kRowBytes);
                };
                // Coarse phases: every fragment read of the k-step is issued before the first MFMA and
                // the MFMAs run as one burst. Left alone, hipcc re-reads two A fragments at a time with a
                // short LDS wait in front of every 8 MFMAs; the two waves of a SIMD then wait and compute
                // in lockstep (SQ_WAIT_ANY 49 %, MFMA pipe 51 % busy at the held clock).
                auto mm = [&]() {
                    __builtin_amdgcn_sched_barrier(0);
                    _Pragma("unroll") for (int i = 0; i < kMS; ++i)
                        _Pragma("unroll") for (int j = 0; j < C::NS; ++j) mma_step<T>(af[i], bf[j], acc[i][j]);
                    __builtin_amdgcn_sched_barrier(0);
                };
#if MLA_CONV_STAMPS == 1
                const bool stamp_on = tap == MLA_STAMP_TAP && chunk_counter == MLA_STAMP_CHUNK && (wave == 0 || wave == 4) && blockIdx.x < 8 && blockIdx.y == 0;
#endif
                if constexpr (!LATE) {
                    MLA_STAMP(0);
#if MLA_CONV_PRIO == 3
                    __builtin_amdgcn_s_setprio(2);             // its first burst outranks the partner's k-step-0 burst (prio 1)
#endif
                    if (!MLA_CONV_DMA_LATE) issue_dma();
                    rd(0);
                    if (MLA_CONV_DMA_LATE) issue_dma();
                    MLA_STAMP(1);
#if MLA_CONV_STAMPS == 1
                    if (stamp_on) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
                    MLA_STAMP(2);
                    mm();
                    MLA_STAMP(3);
#if MLA_CONV_PRIO == 3
                    __builtin_amdgcn_s_setprio(0);             // ... its second burst yields to it
#endif
                    rd(1);
#if MLA_CONV_STAMPS == 1
                    if (stamp_on) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
                    MLA_STAMP(4);
                    mm();
                    MLA_STAMP(5);
                } else {
#if MLA_CONV_PRIO == 3
                    __builtin_amdgcn_s_setprio(3);             // the carried-over burst goes first, uninterrupted
#endif
                    MLA_STAMP(0);
                    if (!MLA_CONV_DMA_LATE) issue_dma();
                    if (tap > 0 || c > 0) mm();    // k-step 1 of the previous tap: fragments were read before the barrier
                    if (MLA_CONV_DMA_LATE) issue_dma();
                    MLA_STAMP(1);
#if MLA_CONV_PRIO == 3
                    __builtin_amdgcn_s_setprio(1);
#endif
                    rd(0);
#if MLA_CONV_STAMPS == 1
                    if (stamp_on) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
                    MLA_STAMP(2);
                    mm();
                    MLA_STAMP(3);
                    rd(1);                         // multiplied after the barrier
                    MLA_STAMP(4);
                    // the reads must have left the slice before the barrier lets the next DMA overwrite it
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    MLA_STAMP(5);
                }
                // The DMA'd slice must have landed before the barrier that publishes it: LDS-DMA is ordered for a ds_read only
                // by the issuing wave's vmcnt followed by a barrier. hipcc emits this wait itself in front of
                // __syncthreads() while a DMA is in flight; it is spelled out so correctness does not rest on that.
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                MLA_STAMP(6);
                __syncthreads();                   // ... and every wave is done with `cur`
                MLA_STAMP(7);
#if MLA_CONV_STAMPS
                if (tap == 8) ++chunk_counter;
#endif
                if constexpr (C::A_DMA) {
                    if (tap == 8 && new_patch) abuf ^= 1;
                } else {
                    if (tap == 8 && new_patch) { // every wave has finished reading the old patch
                        a_write();
                        __syncthreads();
                    }
                }
                par ^= 1;
            }
        }

#if MLA_CONV_STAMPS == 2
        const bool stamp_on = chunk_counter == MLA_STAMP_CHUNK && (wave == 0 || wave == 4) && blockIdx.x < 8 && blockIdx.y == 0;
        MLA_STAMP2(0);
#endif
        if constexpr (LATE) {                      // the last tap's second k-step
            __builtin_amdgcn_sched_barrier(0);
            _Pragma("unroll") for (int i = 0; i < kMS; ++i)
                _Pragma("unroll") for (int j = 0; j < C::NS; ++j) mma_step<T>(af[i], bf[j], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
        }
#if MLA_CONV_STAMPS == 2
        MLA_STAMP2(1);
#endif
        // epilogue: ReLU (+ lane-local 2x2 max-pool; max commutes with the monotone ReLU; the bias is already in the
        // accumulators); one vector store per pixel.
        {
            if (C::POOL) {
                // training forward (mla_conv3x3_train): the pre-pool post-ReLU activation is kept as well -- the pool / ReLU backward
                // routes gradients with it -- so that the pooled layers need no separate max-pool pass over a tensor just written
                if (!C::SPLIT && prepool) {
                    _Pragma("unroll") for (int i = 0; i < kMS; ++i) {
                        const int y = y_tile + l_y0 + i;
                        _Pragma("unroll") for (int e = 0; e < 4; ++e) {
                            float v[C::NS];
                            _Pragma("unroll") for (int j = 0; j < C::NS; ++j) v[j] = fmaxf(acc[i][j][e], 0.f);
                            const int rr = 4 * q + e;
                            const int img = img0 + (C::SEGW == 8 ? wimg + (rr >> 3) : 0);
                            const int x = C::SEGW == 8 ? (rr & 7) : ((wxh * 16) + rr);
                            if (img < n_img) store_vec<T, C::NS>(prepool + ((size_t(img) * C::H + y) * C::W + x) * C::COUT + nb, v);
                        }
                    }
                }
                _Pragma("unroll") for (int ip = 0; ip < kMS / 2; ++ip) {
                    float p0[C::NS], p1[C::NS];
                    _Pragma("unroll") for (int j = 0; j < C::NS; ++j) {
                        const f32x4 u = acc[2 * ip][j], d = acc[2 * ip + 1][j];
                        p0[j] = fmaxf(fmaxf(fmaxf(u.x, u.y), fmaxf(d.x, d.y)), 0.f);
                        p1[j] = fmaxf(fmaxf(fmaxf(u.z, u.w), fmaxf(d.z, d.w)), 0.f);
                    }
                    const int yo = (y_tile + l_y0 + 2 * ip) >> 1;
                    const int img = img0 + (C::SEGW == 8 ? wimg + (q >> 1) : 0);
                    const int xo = C::SEGW == 8 ? 2 * (q & 1) : (((wxh * 16) + 4 * q) >> 1);
                    if (img < n_img) {
                        T* o = out + ((size_t(img) * C::HO + yo) * C::WO + xo) * C::COUT_MEM + nb;
                        store_px<C>(o, p0);
                        store_px<C>(o + C::COUT_MEM, p1);
                    }
                    // training forward, compact form (mla_conv3x3_train_codes): instead of the pre-pool activation (8 bytes per pooled
                    // element in bf16) one BYTE -- the window position of the first maximum in MaxPool2d's order (0,0) (0,1) (1,0) (1,1),
                    // or 4 where the ReLU is off -- which is all the pool / ReLU backward needs from it
                    if constexpr (!C::SPLIT) {
                        if (codes && img < n_img) {
                            auto code_of = [](float a0, float a1, float a2, float a3) -> uint32_t {
                                const bool m1 = a1 > a0;
                                const float b1 = m1 ? a1 : a0;
                                const bool m2 = a2 > b1;
                                const float b2 = m2 ? a2 : b1;
                                const bool m3 = a3 > b2;
                                const float b3 = m3 ? a3 : b2;
                                return b3 > 0.f ? (m3 ? 3u : (m2 ? 2u : (m1 ? 1u : 0u))) : 4u;
                            };
                            uint32_t c0 = 0u, c1 = 0u;
                            _Pragma("unroll") for (int j = 0; j < C::NS; ++j) {
                                const f32x4 u = acc[2 * ip][j], d = acc[2 * ip + 1][j];
                                c0 |= code_of(u.x, u.y, d.x, d.y) << (8 * j);
                                c1 |= code_of(u.z, u.w, d.z, d.w) << (8 * j);
                            }
                            uint8_t* cp = codes + ((size_t(img) * C::HO + yo) * C::WO + xo) * C::COUT + nb;     // the lane's NS consecutive channels
                            if constexpr (C::NS == 4) {
                                *reinterpret_cast<uint32_t*>(cp) = c0;
                                *reinterpret_cast<uint32_t*>(cp + C::COUT) = c1;
                            } else if constexpr (C::NS == 2) {
                                *reinterpret_cast<uint16_t*>(cp) = uint16_t(c0);
                                *reinterpret_cast<uint16_t*>(cp + C::COUT) = uint16_t(c1);
                            } else {
                                cp[0] = uint8_t(c0);
                                cp[C::COUT] = uint8_t(c1);
                            }
                        }
                    }
                }
            } else {
                _Pragma("unroll") for (int i = 0; i < kMS; ++i) {
                    const int y = y_tile + l_y0 + i;
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) {
                        float v[C::NS];
                        _Pragma("unroll") for (int j = 0; j < C::NS; ++j) v[j] = C::ACT ? fmaxf(acc[i][j][e], 0.f) : acc[i][j][e];
                        const int rr = 4 * q + e;
                        const int img = img0 + (C::SEGW == 8 ? wimg + (rr >> 3) : 0);
                        const int x = C::SEGW == 8 ? (rr & 7) : ((wxh * 16) + rr);
                        if (img < n_img) store_px<C>(out + ((size_t(img) * C::H + y) * C::W + x) * C::COUT_MEM + nb, v);
                    }
                }
            }
        }
#if MLA_CONV_STAMPS == 2
        MLA_STAMP2(2);
#endif
        if (!has_next) break;
        tile = next_tile;
        img0 = (tile / C::TILES_Y) * C::IMGS;
        _Pragma("unroll") for (int i = 0; i < kMS; ++i)
            _Pragma("unroll") for (int j = 0; j < C::NS; ++j) acc[i][j] = f32x4{bv[j], bv[j], bv[j], bv[j]};
    }
    };
    if constexpr (C::STAGGER) {
        if (late) run(std::true_type{});
        else run(std::false_type{});
    } else {
        run(std::false_type{});
    }
#if MLA_CONV_STAMPS
    if ((wave == 0 || wave == 4) && lane == 0 && blockIdx.x < 8 && blockIdx.y == 0)
        for (int k = 0; k < 8; ++k) g_conv_stamps[blockIdx.x][wave >> 2][k] = stamps[k];
#endif
}

// ---------------------------------------------------------------- conv1 (Cin = 1) ----------
// x: (N, 96, 64) log-mel examples; w: (64, 1, 3, 3) f32 as stored by the reference; out: pooled
// NHWC (N, 48, 32, 64). K = 9 taps only, so the layer is bound by the 196 KB/clip (bf16) it writes;
// the arithmetic still goes to the matrix cores (a VALU version needed 2 304 FMAs per pooled pixel and
// was VALU-bound at 2.6x the time): bf16 mode pads K to 32 in one v_mfma_f32_16x16x32_bf16 (taps 0..7
// on the q = 0 lanes, tap 8 on q = 1, zeros elsewhere), f32 mode uses three v_mfma_f32_16x16x4_f32.
// Workgroup = 8 input rows x 64 columns of one clip; wave w owns rows 2w, 2w+1 = one pooled row; the A
// operand is gathered from an f32 LDS patch (10 x 66 with halo); pooled + bias + ReLU results are parked
// in LDS and leave as one contiguous 4 KiB (bf16) block per wave.
template <typename T> struct Conv1Frag;
template <> struct Conv1Frag<bf16_t> { u32x4 b[4]; };           // B[k = 8q + e][n = 16 j + r], bf16 pairs
template <> struct Conv1Frag<float> { float b[3][4]; };         // B[k = 4 s + q][n = 16 j + r]

template <typename TIN, typename T, bool SPLIT = false>
__global__ __launch_bounds__(256, 4) void conv1_kernel(const TIN* __restrict__ x, const float* __restrict__ w,
                                                    const float* __restrict__ bias, T* __restrict__ out, int n_img) {
    constexpr int PITCH = 68;                                   // floats per patch row (66 used)
    constexpr int ROW = 64 * int(sizeof(T)) + 16;               // bytes per pooled pixel in the output stage
    __shared__ float sX[20 * PITCH];                            // second half stays 0.0f: target of the padded taps at any shift
    __shared__ __attribute__((aligned(16))) char sOut[4 * 32 * ROW];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 15, q = lane >> 4;
    const int n_tiles = n_img * 12;                              // 8 input rows of one clip per tile

    // patch of one tile (input rows y0-1 .. y0+8, columns -1 .. 64) -> 3 registers per thread
    float pre[3];
    auto patch_load = [&](int tile) {
        const int img = tile / 12, y0 = (tile % 12) * 8;
        _Pragma("unroll") for (int k = 0; k < 3; ++k) {
            const int p = t + 256 * k, yh = p / 66, xh = p % 66;
            const int gy = y0 + yh - 1, gx = xh - 1;
            // unconditional load from a clamped address + select (a conditional load is branched around and waited for one by one)
            const int cy = gy < 0 ? 0 : (gy > 95 ? 95 : gy), cx = gx < 0 ? 0 : (gx > 63 ? 63 : gx);
            const float v = load_elem<TIN>(x + (size_t(img) * 96 + cy) * 64 + cx);
            pre[k] = (p < 660 && gy >= 0 && gy < 96 && gx >= 0 && gx < 64) ? v : 0.f;
        }
    };
    auto patch_write = [&]() {
        _Pragma("unroll") for (int k = 0; k < 3; ++k) {
            const int p = t + 256 * k;
            if (p < 660) sX[(p / 66) * PITCH + p % 66] = pre[k];
        }
    };
    int tile = blockIdx.x;
    patch_load(tile);
    for (int p = t; p < 10 * PITCH; p += 256) sX[10 * PITCH + p] = 0.f;

    // per-lane constants: weight fragments and the patch offsets of "its" taps
    Conv1Frag<T> wf;
    int toff[8];
    if constexpr (sizeof(T) == 2) {
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {
            uint32_t pk[4];
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {
                const int t0 = 8 * q + 2 * e, t1 = t0 + 1;
                const float v0 = t0 < 9 ? w[(4 * r + j) * 9 + t0] : 0.f, v1 = t1 < 9 ? w[(4 * r + j) * 9 + t1] : 0.f;
                pk[e] = pack_bf16x2(v0, v1);
            }
            wf.b[j] = u32x4{pk[0], pk[1], pk[2], pk[3]};
        }
        _Pragma("unroll") for (int e = 0; e < 8; ++e) {
            const int tt = 8 * q + e;
            toff[e] = tt < 9 ? (tt / 3) * PITCH + tt % 3 + r : 10 * PITCH;      // absolute index (zero slot for padding)
        }
    } else {
        _Pragma("unroll") for (int s3 = 0; s3 < 3; ++s3) {
            const int tt = 4 * s3 + q;
            _Pragma("unroll") for (int j = 0; j < 4; ++j) wf.b[s3][j] = tt < 9 ? w[(4 * r + j) * 9 + tt] : 0.f;
            toff[s3] = tt < 9 ? (tt / 3) * PITCH + tt % 3 + r : 10 * PITCH;
        }
    }
    float bj[4];
    _Pragma("unroll") for (int j = 0; j < 4; ++j) bj[j] = bias[4 * r + j];      // MFMA column r of subtile j = output channel 4 r + j:
                                                                                 // a lane's four subtile values are 4 consecutive channels

    // persistent over tiles: weights / offsets load once; the next tile's patch is in flight during the MFMAs
    char* stage = sOut + wave * 32 * ROW;
    for (; tile < n_tiles; tile += gridDim.x) {
    const int img = tile / 12, y0 = (tile % 12) * 8;
    patch_write();
    __syncthreads();
    if (tile + int(gridDim.x) < n_tiles) patch_load(tile + int(gridDim.x));
    _Pragma("unroll") for (int seg = 0; seg < 4; ++seg) {
        f32x4 acc[2][4];
        _Pragma("unroll") for (int i = 0; i < 2; ++i)
            _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {
            // tap (ky, kx) of pixel (row, x) sits at sX[(row + ky) * PITCH + x + kx]; padded taps read the zero slot
            const int shift = (2 * wave + i) * PITCH + 16 * seg;
            if constexpr (sizeof(T) == 2) {
                float v[8];
                _Pragma("unroll") for (int e = 0; e < 8; ++e) v[e] = sX[toff[e] + shift];
                const u32x4 a = u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
                _Pragma("unroll") for (int j = 0; j < 4; ++j) mma_step<bf16_t>(a, wf.b[j], acc[i][j]);
            } else {
                _Pragma("unroll") for (int s3 = 0; s3 < 3; ++s3) {
                    const float a = sX[toff[s3] + shift];
                    _Pragma("unroll") for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wf.b[s3][j], acc[i][j], 0, 0, 0);
                }
            }
        }
        // 2x2 pool (lane-local), bias, ReLU -> LDS stage [pooled x][64 ch]: the lane's four channels 4r .. 4r+3 as one piece
        {
            float p0[4], p1[4];
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {
                const f32x4 u = acc[0][j], d = acc[1][j];
                p0[j] = fmaxf(fmaxf(fmaxf(u.x, u.y), fmaxf(d.x, d.y)) + bj[j], 0.f);
                p1[j] = fmaxf(fmaxf(fmaxf(u.z, u.w), fmaxf(d.z, d.w)) + bj[j], 0.f);
            }
            const int xo = 8 * seg + 2 * q;
            if constexpr (SPLIT) {        // f32 arithmetic, [hi(64) | lo(64)] bf16 per pooled pixel = the f32 row's 256 bytes
                static_assert(!SPLIT || sizeof(T) == 4, "split output is produced by the exact-f32 path");
                float h0[4], h1[4], l0[4], l1[4];
                _Pragma("unroll") for (int j = 0; j < 4; ++j) {
                    h0[j] = bf2f(f2bf(p0[j])); l0[j] = p0[j] - h0[j];
                    h1[j] = bf2f(f2bf(p1[j])); l1[j] = p1[j] - h1[j];
                }
                bf16_t* s0 = reinterpret_cast<bf16_t*>(stage + xo * ROW) + 4 * r;
                bf16_t* s1 = reinterpret_cast<bf16_t*>(stage + (xo + 1) * ROW) + 4 * r;
                store_vec<bf16_t, 4>(s0, h0); store_vec<bf16_t, 4>(s0 + 64, l0);
                store_vec<bf16_t, 4>(s1, h1); store_vec<bf16_t, 4>(s1 + 64, l1);
            } else {
                store_vec<T, 4>(reinterpret_cast<T*>(stage + xo * ROW) + 4 * r, p0);
                store_vec<T, 4>(reinterpret_cast<T*>(stage + (xo + 1) * ROW) + 4 * r, p1);
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the stage rows belong to this wave only:
    __builtin_amdgcn_wave_barrier();                            // LDS operations of one wave execute in order
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    constexpr int PIECES = 64 * int(sizeof(T)) / 16;            // 16-B pieces per pooled pixel
    char* dst = reinterpret_cast<char*>(out + ((size_t(img) * 48 + y0 / 2 + wave) * 32) * 64);
    _Pragma("unroll") for (int it = 0; it < PIECES / 2; ++it) {
        const int piece = it * 64 + lane, p = piece / PIECES, c = piece % PIECES;
        *reinterpret_cast<u32x4*>(dst + size_t(piece) * 16) = *reinterpret_cast<const u32x4*>(stage + p * ROW + c * 16);
    }
    __syncthreads();                                            // every wave is done with sX before the next patch lands
    }
}

// ---- conv1, bf16 output: the layer as a K = 16 GEMM over 4 x 4 INPUT PATCHES (round 2) --------------------------------------
// The four pre-pool outputs of a pooled pixel read the same 4 x 4 input patch (rows 2py-1 .. 2py+2, columns 2px-1 .. 2px+2), so
//   pre[pos][ch] = sum_{k < 16} W'[pos][ch][k] * patch[k],   W'[(dy,dx)][ch][4 (dy+ky) + (dx+kx)] = w[ch][ky][kx], 0 elsewhere,
// which is exactly one v_mfma_f32_32x32x16_bf16 per (position, 32 channels) with the WEIGHTS as the A operand (held in registers
// for the workgroup's life) and 32 pooled pixels of one pooled row as the B operand: a lane's fragment is two patch rows of its
// pixel = two ds_read2_b32 (the gather form above needs 8 ds_read_b32 per 16 PRE-pool pixels: 32x the LDS instructions), the 2 x 2
// max-pool is an elementwise max over the four positions' accumulators, and with the weight rows permuted (MFMA row rho holds
// channel 16 ((rho >> 2) & 1) + (rho & 3) + 4 (rho >> 3) of its 32) a lane ends up with 16 CONSECUTIVE channels of its pixel.
// 8 MFMAs per 32 pooled pixels x 64 channels instead of 32 x 16x16x32. Accumulators start at the bias (max commutes with + b).
template <typename TIN>
__global__ __launch_bounds__(256, MLA_CONV1_WAVES) void conv1_patch_kernel(const TIN* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, bf16_t* __restrict__ out, int n_img) {
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    constexpr int TR = MLA_CONV1_ROWS;                          // input rows per tile (TR / 2 pooled rows, TR / 8 per wave)
    constexpr int TPC = 96 / TR;                                // tiles per clip
    constexpr int NP = (TR + 2) * 66;                           // staged input elements per tile
    constexpr int NL = (NP + 255) / 256;                        // ... per thread
    constexpr int PITCH = 68;                                   // bf16 per staged input row: 66 used, element (gy, gx) at column gx + 1
    constexpr int ROW = 64 * 2 + 16;                            // bytes per pooled pixel in the output stage
    __shared__ __attribute__((aligned(16))) uint16_t sX[(TR + 2) * PITCH];
    __shared__ __attribute__((aligned(16))) char sOut[4 * 32 * ROW];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, px = lane & 31, h = lane >> 5;
    const int n_tiles = n_img * TPC;

    // A operand: W'[pos][channel(rho)][8 h + j] for rho = lane & 31, per position and 32-channel tile
    bf16x8 wa[4][2];
    f32x16 binit[2];
    {
        const int rho = px;
        const int chl = 16 * ((rho >> 2) & 1) + (rho & 3) + 4 * (rho >> 3);
        _Pragma("unroll") for (int pos = 0; pos < 4; ++pos) {
            const int dy = pos >> 1, dx = pos & 1;
            _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) {
                const int ch = mt * 32 + chl;
                uint32_t pk[4];
                _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) {
                    float v[2];
                    _Pragma("unroll") for (int e = 0; e < 2; ++e) {
                        const int k = 8 * h + 2 * jj + e, i = k >> 2, c = k & 3;          // patch row i, column c
                        const int ky = i - dy, kx = c - dx;
                        v[e] = (ky >= 0 && ky < 3 && kx >= 0 && kx < 3) ? w[ch * 9 + ky * 3 + kx] : 0.f;
                    }
                    pk[jj] = pack_bf16x2(v[0], v[1]);
                }
                wa[pos][mt] = __builtin_bit_cast(bf16x8, u32x4{pk[0], pk[1], pk[2], pk[3]});
            }
        }
        // C/D: lane (col = pixel, hi = h) register reg <-> MFMA row (reg & 3) + 8 (reg >> 2) + 4 hi <-> channel 16 hi + reg of the tile
        _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)
            _Pragma("unroll") for (int reg = 0; reg < 16; ++reg) binit[mt][reg] = bias[mt * 32 + 16 * h + reg];
    }

    // input rows y0-1 .. y0+TR, columns -1 .. 64 of one tile -> NL registers per thread, then LDS (bf16)
    float pre[NL];
    auto patch_load = [&](int tile) {
        const int img = tile / TPC, y0 = (tile % TPC) * TR;
        _Pragma("unroll") for (int k = 0; k < NL; ++k) {
            const int p = t + 256 * k, yh = p / 66, xh = p % 66;
            const int gy = y0 + yh - 1, gx = xh - 1;
            // UNCONDITIONAL load from a clamped address, then a select: a load under a lane-dependent condition makes hipcc branch
            // around it and wait s_waitcnt vmcnt(0) right behind it -- every element a serial memory round trip that also waits
            // for all outstanding stores (the round-1 kernel did exactly that: 3.6 TB/s of output with the arithmetic removed)
            const int cy = gy < 0 ? 0 : (gy > 95 ? 95 : gy), cx = gx < 0 ? 0 : (gx > 63 ? 63 : gx);
            const float v = load_elem<TIN>(x + (size_t(img) * 96 + cy) * 64 + cx);
            pre[k] = (p < NP && gy >= 0 && gy < 96 && gx >= 0 && gx < 64) ? v : 0.f;
        }
    };
    auto patch_write = [&]() {
        _Pragma("unroll") for (int k = 0; k < NL; ++k) {
            const int p = t + 256 * k;
            if (p < NP) sX[(p / 66) * PITCH + p % 66] = f2bf(pre[k]);
        }
    };
    // The two barriers of a tile protect LDS only. __syncthreads() would also wait for every outstanding global STORE
    // (s_waitcnt vmcnt(0): ~3-4 us under a full write load) twice per tile -- the round-1 kernel ran at 3.6 TB/s of output for that
    // reason alone (stores only, no arithmetic: 0.556 ms; a plain fill of the same 2 GB: 0.29 ms). Raw barrier + LDS counter only.
    auto lds_barrier = [&]() {
#if MLA_CONV1_RAW_BARRIER
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#else
        __syncthreads();
#endif
    };
    int tile = blockIdx.x;
    if (tile < n_tiles) patch_load(tile);
    char* stage = sOut + wave * 32 * ROW;
    for (; tile < n_tiles; tile += gridDim.x) {
        const int img = tile / TPC, y0 = (tile % TPC) * TR;
        patch_write();
        lds_barrier();
        if (tile + int(gridDim.x) < n_tiles) patch_load(tile + int(gridDim.x));
        _Pragma("unroll") for (int rr = 0; rr < TR / 8; ++rr) {
            const int prow = wave * (TR / 8) + rr;                                  // pooled row of the tile owned by this wave
            // B operand: patch rows 2h, 2h+1 of its pixel (local input rows 2 prow + 2h, + 1), columns 2 px .. 2 px + 3 of sX
            const uint32_t* r0 = reinterpret_cast<const uint32_t*>(sX + (2 * prow + 2 * h) * PITCH) + px;       // 2 px bf16 = px dwords
            const uint32_t* r1 = reinterpret_cast<const uint32_t*>(sX + (2 * prow + 2 * h + 1) * PITCH) + px;
            const bf16x8 bfrag = __builtin_bit_cast(bf16x8, u32x4{r0[0], r0[1], r1[0], r1[1]});
            _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) {
                f32x16 m = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[0][mt], bfrag, binit[mt], 0, 0, 0);
                _Pragma("unroll") for (int pos = 1; pos < 4; ++pos) {
                    const f32x16 a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[pos][mt], bfrag, binit[mt], 0, 0, 0);
                    _Pragma("unroll") for (int reg = 0; reg < 16; ++reg) m[reg] = fmaxf(m[reg], a[reg]);
                }
                uint32_t pk[8];
                _Pragma("unroll") for (int e = 0; e < 8; ++e) pk[e] = pack_bf16x2(fmaxf(m[2 * e], 0.f), fmaxf(m[2 * e + 1], 0.f));
                char* dst = stage + px * ROW + (mt * 32 + 16 * h) * 2;              // 16 consecutive channels of pixel px
                *reinterpret_cast<u32x4*>(dst) = u32x4{pk[0], pk[1], pk[2], pk[3]};
                *reinterpret_cast<u32x4*>(dst + 16) = u32x4{pk[4], pk[5], pk[6], pk[7]};
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the stage rows belong to this wave only:
            __builtin_amdgcn_wave_barrier();                            // LDS operations of one wave execute in order
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            char* gdst = reinterpret_cast<char*>(out + ((size_t(img) * 48 + y0 / 2 + prow) * 32) * 64);
            _Pragma("unroll") for (int it = 0; it < 4; ++it) {                      // 32 pixels x 128 B = 4 KiB contiguous per wave
                const int piece = it * 64 + lane, p = piece >> 3, c = piece & 7;
#if MLA_CONV_NT_STORE
                __builtin_nontemporal_store(*reinterpret_cast<const u32x4*>(stage + p * ROW + c * 16), reinterpret_cast<u32x4*>(gdst + size_t(piece) * 16));
#else
                *reinterpret_cast<u32x4*>(gdst + size_t(piece) * 16) = *reinterpret_cast<const u32x4*>(stage + p * ROW + c * 16);
#endif
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the reads above precede the next row's stage writes
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        lds_barrier();                                              // every wave is done with sX before the next patch lands
    }
}

// ------------------------------------------------------------- weight re-layout ------------
// (Cout, Cin, 3, 3) f32 (state_dict layout, vggish.py:113) -> (Cout, 9, Cin) in the compute type
template <typename T>
__global__ void repack_kernel(const float* __restrict__ w, T* __restrict__ out, int cout, int cin) {
    const int64_t total = int64_t(cout) * 9 * cin;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += int64_t(gridDim.x) * blockDim.x) {
        const int c = int(i % cin);
        const int tap = int((i / cin) % 9);
        const int o = int(i / (int64_t(cin) * 9));
        store_elem<T>(out + i, w[(int64_t(o) * cin + c) * 9 + tap]);
    }
}

template <typename T>
__global__ void convert_kernel(const float* __restrict__ in, T* __restrict__ out, int64_t n) {
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x)
        store_elem<T>(out + i, in[i]);
}

__global__ void widen_kernel(const bf16_t* __restrict__ in, float* __restrict__ out, int64_t n) {
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x)
        out[i] = bf2f(in[i].bits);
}

inline bool wide_tiles() {
    const char* e = getenv("MLA_CONV_TILE");
    return e && e[0] == 'w';
}

template <typename C>
int launch_conv(const void* in, const void* w, const float* bias, void* out, int64_t n_img, hipStream_t s, void* prepool = nullptr,
                uint8_t* codes = nullptr) {
    using T = typename C::elem;
    auto kern = conv3x3_kernel<C>;
    MLA_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   C::LDS_BYTES));
    const int64_t tiles = ((n_img + C::IMGS - 1) / C::IMGS) * C::TILES_Y;
    MLA_REQUIRE(tiles <= 0x7fffffff, MLA_E_SHAPE, "too many tiles");
    // persistent grid: one workgroup per CU and N-tile (two where LDS/VGPRs allow), a multiple of TILES_Y
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    constexpr int n_tiles_n = C::COUT / C::BN;
    constexpr int per_cu = (C::LDS_BYTES * 2 <= 160 * 1024 && C::MIN_WAVES >= 4) ? 2 : 1;
    int64_t gx = int64_t(cus) * per_cu / n_tiles_n;
    gx = gx / C::TILES_Y * C::TILES_Y;
    if (gx < C::TILES_Y) gx = C::TILES_Y;
    if (gx > tiles || !C::PERSIST) gx = tiles;
    hipLaunchKernelGGL(kern, dim3(unsigned(gx), n_tiles_n), dim3(kThreads), C::LDS_BYTES, s,
                       static_cast<const T*>(in), static_cast<const T*>(w), bias, static_cast<T*>(out), int(n_img), int(tiles), static_cast<T*>(prepool), codes);
    MLA_LAUNCH_OK("conv3x3_kernel");
    return MLA_OK;
}

template <typename T>
int conv_layer(int layer, const void* in, const void* w, const float* bias, void* out, int64_t n, hipStream_t s) {
    // MLA_CONV_TILE=wide selects the 192-pixel tiles at run time: same products in the same order per accumulator, i.e. bit-identical
    // results (tests/test_model_gpu.py checks that), for A/B timing and as the cross-check of the tall tiles' index arithmetic
    if constexpr (sizeof(T) == 2 && MLA_CONV_TALL) {     // bf16: tall tiles (384 pixels x 128 channels)
        if (!wide_tiles()) switch (layer) {
            case 2: return launch_conv<Cfg<T, 64, 128, 48, 32, true, 4, true, false, 4>>(in, w, bias, out, n, s);
            case 3: return launch_conv<Cfg<T, 128, 256, 24, 16, false, 4, true, false, 4>>(in, w, bias, out, n, s);
            case 4: return launch_conv<Cfg<T, 256, 256, 24, 16, true, 4, true, false, 4>>(in, w, bias, out, n, s);
#if MLA_CONV_TALL >= 2
            case 5: return launch_conv<Cfg<T, 256, 512, 12, 8, false, 4, true, false, 4>>(in, w, bias, out, n, s);
            case 6: return launch_conv<Cfg<T, 512, 512, 12, 8, true, 4, true, false, 4>>(in, w, bias, out, n, s);
#endif
        }
    }
    switch (layer) {                                     //        Cin Cout  H   W  pool NS
        case 2: return launch_conv<Cfg<T, 64, 128, 48, 32, true, 2>>(in, w, bias, out, n, s);
        case 3: return launch_conv<Cfg<T, 128, 256, 24, 16, false, 4>>(in, w, bias, out, n, s);
        case 4: return launch_conv<Cfg<T, 256, 256, 24, 16, true, 4>>(in, w, bias, out, n, s);
        case 5: return launch_conv<Cfg<T, 256, 512, 12, 8, false, 4>>(in, w, bias, out, n, s);
        case 6: return launch_conv<Cfg<T, 512, 512, 12, 8, true, 4>>(in, w, bias, out, n, s);
    }
    return mla::fail(MLA_E_SHAPE, "conv layer %d is not one of VGGish conv2..conv6", layer);
}

int conv_layer_split(int layer, const void* in, const void* w, const float* bias, void* out, int64_t n, hipStream_t s) {
    if (MLA_CONV_TALL_SPLIT && !wide_tiles()) switch (layer) {
        case 2: return launch_conv<Cfg<bf16_t, 64, 128, 48, 32, true, 4, true, true, 4>>(in, w, bias, out, n, s);
        case 3: return launch_conv<Cfg<bf16_t, 128, 256, 24, 16, false, 4, true, true, 4>>(in, w, bias, out, n, s);
        case 4: return launch_conv<Cfg<bf16_t, 256, 256, 24, 16, true, 4, true, true, 4>>(in, w, bias, out, n, s);
        case 5: return launch_conv<Cfg<bf16_t, 256, 512, 12, 8, false, 4, true, true, 4>>(in, w, bias, out, n, s);
        case 6: return launch_conv<Cfg<bf16_t, 512, 512, 12, 8, true, 4, true, true, 4>>(in, w, bias, out, n, s);
    }
    switch (layer) {
        case 2: return launch_conv<Cfg<bf16_t, 64, 128, 48, 32, true, 2, true, true>>(in, w, bias, out, n, s);
        case 3: return launch_conv<Cfg<bf16_t, 128, 256, 24, 16, false, 4, true, true>>(in, w, bias, out, n, s);
        case 4: return launch_conv<Cfg<bf16_t, 256, 256, 24, 16, true, 4, true, true>>(in, w, bias, out, n, s);
        case 5: return launch_conv<Cfg<bf16_t, 256, 512, 12, 8, false, 4, true, true>>(in, w, bias, out, n, s);
        case 6: return launch_conv<Cfg<bf16_t, 512, 512, 12, 8, true, 4, true, true>>(in, w, bias, out, n, s);
    }
    return mla::fail(MLA_E_SHAPE, "conv layer %d is not one of VGGish conv2..conv6", layer);
}

// (Cout, Cin, 3, 3) f32 -> (Cin, 9, Cout) with the taps flipped: the weights of the transposed
// convolution that maps dZ (N,H,W,Cout) to dA (N,H,W,Cin):  W'[ci][8 - tap][co] = W[co][ci][tap]
__global__ void repack_dgrad_kernel(const float* __restrict__ w, float* __restrict__ out, int cout, int cin) {
    const int64_t total = int64_t(cout) * 9 * cin;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += int64_t(gridDim.x) * blockDim.x) {
        const int co = int(i % cout);
        const int tap = int((i / cout) % 9);
        const int ci = int(i / (int64_t(cout) * 9));
        out[i] = w[(int64_t(co) * cin + ci) * 9 + (8 - tap)];
    }
}

// every compiled (H, W, Cin, Cout, pool, act) combination of the generic entry: VGGish forward with and without the
// fused pool (training keeps the pre-pool activations) and the five dgrad shapes; f32 (exact) and bf16
template <typename T>
int conv_generic(const void* in, const void* w, const float* bias, void* out, int64_t n, int H, int W, int cin, int cout,
                 bool pool, bool act, hipStream_t s, void* prepool = nullptr, uint8_t* codes = nullptr) {
#define MLA_CONV_CASE(CI, CO, HH, WW, PO, NS_, AC)                                                             \
    if (cin == CI && cout == CO && H == HH && W == WW && pool == PO && act == AC)                              \
        return launch_conv<Cfg<T, CI, CO, HH, WW, PO, NS_, AC>>(in, w, bias, out, n, s, prepool, codes);
    // Cout >= 128: bf16 runs the tall tile (384 pixels x 128 channels, NS = 4), f32 the wide one with NS_ as given
#define MLA_CONV_CASE_TALL(CI, CO, HH, WW, PO, NS_, AC)                                                        \
    if (cin == CI && cout == CO && H == HH && W == WW && pool == PO && act == AC) {                            \
        if constexpr (sizeof(T) == 2 && MLA_CONV_TALL)                                                         \
            if (!wide_tiles())                                                                                 \
                return launch_conv<Cfg<T, CI, CO, HH, WW, PO, 4, AC, false, 4>>(in, w, bias, out, n, s, prepool, codes);  \
        return launch_conv<Cfg<T, CI, CO, HH, WW, PO, NS_, AC>>(in, w, bias, out, n, s, prepool, codes);              \
    }
#if MLA_CONV_TALL >= 2
#define MLA_CONV_CASE_TALL8 MLA_CONV_CASE_TALL
#else
#define MLA_CONV_CASE_TALL8 MLA_CONV_CASE
#endif
    MLA_CONV_CASE_TALL(64, 128, 48, 32, true, 2, true)
    MLA_CONV_CASE_TALL(128, 256, 24, 16, false, 4, true)
    MLA_CONV_CASE_TALL(256, 256, 24, 16, true, 4, true)
    MLA_CONV_CASE_TALL8(256, 512, 12, 8, false, 4, true)
    MLA_CONV_CASE_TALL8(512, 512, 12, 8, true, 4, true)
    MLA_CONV_CASE_TALL(64, 128, 48, 32, false, 2, true)   // training forward: pre-pool activations kept
    MLA_CONV_CASE_TALL(256, 256, 24, 16, false, 4, true)
    MLA_CONV_CASE_TALL8(512, 512, 12, 8, false, 4, true)
    MLA_CONV_CASE_TALL8(512, 512, 12, 8, false, 4, false)       // dgrad conv6
    MLA_CONV_CASE_TALL8(512, 256, 12, 8, false, 4, false)       // dgrad conv5
    MLA_CONV_CASE_TALL(256, 256, 24, 16, false, 4, false) // dgrad conv4
    MLA_CONV_CASE_TALL(256, 128, 24, 16, false, 2, false) // dgrad conv3
#if MLA_CONV_DGRAD2 >= 1
    // dgrad conv2 (128 -> 64 @48x32), bf16: the tall tile 384 pixels x 64 channels (waves 4 x 2, NS = 2), two workgroups per CU
    if (cin == 128 && cout == 64 && H == 48 && W == 32 && !pool && !act) {
        if constexpr (sizeof(T) == 2) return launch_conv<Cfg<T, 128, 64, 48, 32, false, 2, false, false, 4>>(in, w, bias, out, n, s, prepool, codes);
    }
#endif
    MLA_CONV_CASE(128, 64, 48, 32, false, 1, false)       // dgrad conv2
#undef MLA_CONV_CASE_TALL
#undef MLA_CONV_CASE_TALL8
#undef MLA_CONV_CASE
    return mla::fail(MLA_E_SHAPE, "conv3x3 %dx%d %d->%d pool=%d act=%d is not compiled", H, W, cin, cout, int(pool), int(act));
}

}  // namespace

#if MLA_CONV_STAMPS
extern "C" int mla_debug_conv_stamps(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_conv_stamps), sizeof(g_conv_stamps)) == hipSuccess ? 0 : -4;
}
#endif

extern "C" int mla_conv3x3(const void* in, const void* w_packed, const float* bias, void* out, int64_t n, int H, int W,
                           int cin, int cout, int pool, int act, int dtype, mla_stream_t stream) {
    MLA_REQUIRE(n >= 0, MLA_E_ARG, "n %lld", (long long)n);
    if (n == 0) return MLA_OK;
    MLA_REQUIRE(in && w_packed && out && (bias || !act), MLA_E_ARG, "null conv buffers");
    MLA_REQUIRE(mla::aligned(in, 16) && mla::aligned(w_packed, 16), MLA_E_ARG, "conv buffers must be 16-byte aligned");
    MLA_REQUIRE(dtype == MLA_F32 || dtype == MLA_BF16, MLA_E_DTYPE, "the generic conv entry (training / dgrad) is compiled for f32 and bf16");
    if (dtype == MLA_BF16)
        return conv_generic<bf16_t>(in, w_packed, bias, out, n, H, W, cin, cout, pool != 0, act != 0, static_cast<hipStream_t>(stream));
    return conv_generic<float>(in, w_packed, bias, out, n, H, W, cin, cout, pool != 0, act != 0, static_cast<hipStream_t>(stream));
}

extern "C" int mla_conv3x3_train(const void* in, const void* w_packed, const float* bias, void* out_prepool, void* out_pooled, int64_t n,
                                 int H, int W, int cin, int cout, int dtype, mla_stream_t stream) {
    MLA_REQUIRE(n >= 0, MLA_E_ARG, "n %lld", (long long)n);
    if (n == 0) return MLA_OK;
    MLA_REQUIRE(in && w_packed && bias && out_prepool && out_pooled, MLA_E_ARG, "null conv buffers");
    MLA_REQUIRE(mla::aligned(in, 16) && mla::aligned(w_packed, 16) && mla::aligned(out_prepool, 16), MLA_E_ARG, "conv buffers must be 16-byte aligned");
    MLA_REQUIRE(dtype == MLA_F32 || dtype == MLA_BF16, MLA_E_DTYPE, "conv3x3_train dtype %d", dtype);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == MLA_BF16) return conv_generic<bf16_t>(in, w_packed, bias, out_pooled, n, H, W, cin, cout, true, true, s, out_prepool);
    return conv_generic<float>(in, w_packed, bias, out_pooled, n, H, W, cin, cout, true, true, s, out_prepool);
}

extern "C" int mla_conv3x3_train_codes(const void* in, const void* w_packed, const float* bias, void* out_codes, void* out_pooled, int64_t n,
                                       int H, int W, int cin, int cout, int dtype, mla_stream_t stream) {
    MLA_REQUIRE(n >= 0, MLA_E_ARG, "n %lld", (long long)n);
    if (n == 0) return MLA_OK;
    MLA_REQUIRE(in && w_packed && bias && out_codes && out_pooled, MLA_E_ARG, "null conv buffers");
    MLA_REQUIRE(mla::aligned(in, 16) && mla::aligned(w_packed, 16) && mla::aligned(out_codes, 4), MLA_E_ARG, "conv buffers must be 16-byte aligned (codes: 4)");
    MLA_REQUIRE(dtype == MLA_F32 || dtype == MLA_BF16, MLA_E_DTYPE, "conv3x3_train_codes dtype %d", dtype);
    hipStream_t s = static_cast<hipStream_t>(stream);
    uint8_t* codes = static_cast<uint8_t*>(out_codes);
    if (dtype == MLA_BF16) return conv_generic<bf16_t>(in, w_packed, bias, out_pooled, n, H, W, cin, cout, true, true, s, nullptr, codes);
    return conv_generic<float>(in, w_packed, bias, out_pooled, n, H, W, cin, cout, true, true, s, nullptr, codes);
}

extern "C" int mla_conv_repack_dgrad(const float* w_oihw, int64_t cout, int64_t cin, float* out, mla_stream_t stream) {
    MLA_REQUIRE(w_oihw && out && cout > 0 && cin > 0, MLA_E_ARG, "bad repack arguments");
    const int64_t total = cout * 9 * cin;
    const unsigned grid = unsigned((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(repack_dgrad_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), w_oihw, out, int(cout), int(cin));
    MLA_LAUNCH_OK("repack_dgrad_kernel");
    return MLA_OK;
}

extern "C" int mla_conv_repack_weights(const float* w_oihw, int64_t cout, int64_t cin, void* out, int dtype,
                                       mla_stream_t stream) {
    MLA_REQUIRE(w_oihw && out && cout > 0 && cin > 0, MLA_E_ARG, "bad repack arguments");
    MLA_REQUIRE(dtype == MLA_F32 || dtype == MLA_BF16, MLA_E_DTYPE, "dtype %d", dtype);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t total = cout * 9 * cin;
    const unsigned grid = unsigned((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (dtype == MLA_F32)
        hipLaunchKernelGGL(repack_kernel<float>, dim3(grid), dim3(256), 0, s, w_oihw, static_cast<float*>(out), int(cout), int(cin));
    else
        hipLaunchKernelGGL(repack_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, w_oihw, static_cast<bf16_t*>(out), int(cout), int(cin));
    MLA_LAUNCH_OK("repack_kernel");
    return MLA_OK;
}

extern "C" int mla_convert_f32(const float* in, void* out, int64_t n, int dtype, mla_stream_t stream) {
    MLA_REQUIRE(in && out && n >= 0, MLA_E_ARG, "bad convert arguments");
    MLA_REQUIRE(dtype == MLA_BF16, MLA_E_DTYPE, "convert target must be bf16 (got %d)", dtype);
    if (n == 0) return MLA_OK;
    const unsigned grid = unsigned((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(convert_kernel<bf16_t>, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), in,
                       static_cast<bf16_t*>(out), n);
    MLA_LAUNCH_OK("convert_kernel");
    return MLA_OK;
}

// f32 (rows, cols) -> bf16 planes per segment of `seg` columns: [hi | lo] (copies = 2, activations) or [hi | lo | hi]
// (copies = 3, weights of the bf16x3 mode); out row pitch ld_out >= copies * cols.
__global__ void split_kernel(const float* __restrict__ in, int64_t rows, int64_t cols, int64_t ld_in, bf16_t* __restrict__ out,
                             int64_t ld_out, int64_t seg, int copies) {
    const int64_t total = rows * cols;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += int64_t(gridDim.x) * blockDim.x) {
        const int64_t r = i / cols, c = i - r * cols, sidx = c / seg, within = c - sidx * seg;
        const float x = in[r * ld_in + c];
        const float hi = bf2f(f2bf(x));
        bf16_t* o = out + r * ld_out + sidx * copies * seg + within;
        o[0].bits = f2bf(hi);
        o[seg].bits = f2bf(x - hi);
        if (copies == 3) o[2 * seg].bits = f2bf(hi);
    }
}

extern "C" int mla_split_bf16x3(const float* in, int64_t rows, int64_t cols, int64_t ld_in, void* out, int64_t ld_out,
                                int64_t seg, int copies, mla_stream_t stream) {
    MLA_REQUIRE(rows >= 0 && cols > 0 && seg > 0 && cols % seg == 0 && (copies == 2 || copies == 3) && ld_in >= cols &&
                ld_out >= copies * cols, MLA_E_ARG, "bad split arguments (rows %lld cols %lld seg %lld copies %d)",
                (long long)rows, (long long)cols, (long long)seg, copies);
    if (rows == 0) return MLA_OK;
    MLA_REQUIRE(in && out, MLA_E_ARG, "null split buffers");
    const int64_t total = rows * cols;
    const unsigned grid = unsigned((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(split_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), in, rows, cols, ld_in,
                       static_cast<bf16_t*>(out), ld_out, seg, copies);
    MLA_LAUNCH_OK("split_kernel");
    return MLA_OK;
}

__global__ void merge_kernel(const bf16_t* __restrict__ in, int64_t rows, int64_t cols, int64_t ld_in, int64_t seg,
                             float* __restrict__ out) {
    const int64_t total = rows * cols;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += int64_t(gridDim.x) * blockDim.x) {
        const int64_t r = i / cols, c = i - r * cols, sidx = c / seg, within = c - sidx * seg;
        const bf16_t* p = in + r * ld_in + sidx * 2 * seg + within;
        out[i] = bf2f(p[0].bits) + bf2f(p[seg].bits);
    }
}

extern "C" int mla_merge_bf16x3(const void* in, int64_t rows, int64_t cols, int64_t ld_in, int64_t seg, float* out,
                                mla_stream_t stream) {
    MLA_REQUIRE(rows >= 0 && cols > 0 && seg > 0 && cols % seg == 0 && ld_in >= 2 * cols, MLA_E_ARG, "bad merge arguments");
    if (rows == 0) return MLA_OK;
    MLA_REQUIRE(in && out, MLA_E_ARG, "null merge buffers");
    const int64_t total = rows * cols;
    const unsigned grid = unsigned((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(merge_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const bf16_t*>(in),
                       rows, cols, ld_in, seg, out);
    MLA_LAUNCH_OK("merge_kernel");
    return MLA_OK;
}

extern "C" int mla_convert_bf16_to_f32(const void* in, float* out, int64_t n, mla_stream_t stream) {
    MLA_REQUIRE(in && out && n >= 0, MLA_E_ARG, "bad convert arguments");
    if (n == 0) return MLA_OK;
    const unsigned grid = unsigned((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(widen_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const bf16_t*>(in), out, n);
    MLA_LAUNCH_OK("widen_kernel");
    return MLA_OK;
}

extern "C" int mla_vggish_conv1(const void* x, int x_dtype, int64_t n, const float* w, const float* bias, void* out,
                                int dtype, mla_stream_t stream) {
    MLA_REQUIRE(n >= 0, MLA_E_ARG, "n %lld", (long long)n);
    if (n == 0) return MLA_OK;
    MLA_REQUIRE(x && w && bias && out && mla::aligned(out, 16), MLA_E_ARG, "null / misaligned conv1 buffers");
    MLA_REQUIRE((x_dtype == MLA_F32 || x_dtype == MLA_BF16) && (dtype == MLA_F32 || dtype == MLA_BF16 || dtype == MLA_BF16X3),
                MLA_E_DTYPE, "conv1 dtypes %d -> %d", x_dtype, dtype);
    MLA_REQUIRE(dtype != MLA_BF16X3 || x_dtype == MLA_F32, MLA_E_DTYPE, "the split output is computed from float32 examples");
    MLA_REQUIRE(n * 12 <= 0x7fffffff, MLA_E_SHAPE, "batch too large");
    int dev1 = 0, cus1 = 256;
    if (hipGetDevice(&dev1) == hipSuccess) hipDeviceGetAttribute(&cus1, hipDeviceAttributeMultiprocessorCount, dev1);
    const int64_t blocks = n * 12 < int64_t(cus1) * 4 ? n * 12 : int64_t(cus1) * 4;      // persistent: 4 workgroups per CU (VGPR-limited)
    const int n_pix = int(n);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 g{unsigned(blocks)}, b{256};
    if (dtype == MLA_BF16X3)
        hipLaunchKernelGGL((conv1_kernel<float, float, true>), g, b, 0, s, static_cast<const float*>(x), w, bias, static_cast<float*>(out), n_pix);
    else if (x_dtype == MLA_F32 && dtype == MLA_F32)
        hipLaunchKernelGGL((conv1_kernel<float, float>), g, b, 0, s, static_cast<const float*>(x), w, bias, static_cast<float*>(out), n_pix);
    else if (dtype == MLA_BF16 && !MLA_CONV1_GATHER) {
        const int64_t tiles = n * (96 / MLA_CONV1_ROWS), slots = int64_t(cus1) * MLA_CONV1_WAVES;
        const dim3 gp{unsigned(tiles < slots ? tiles : slots)};
        if (x_dtype == MLA_F32)
            hipLaunchKernelGGL((conv1_patch_kernel<float>), gp, b, 0, s, static_cast<const float*>(x), w, bias, static_cast<bf16_t*>(out), n_pix);
        else
            hipLaunchKernelGGL((conv1_patch_kernel<bf16_t>), gp, b, 0, s, static_cast<const bf16_t*>(x), w, bias, static_cast<bf16_t*>(out), n_pix);
    }
    else if (x_dtype == MLA_F32)
        hipLaunchKernelGGL((conv1_kernel<float, bf16_t>), g, b, 0, s, static_cast<const float*>(x), w, bias, static_cast<bf16_t*>(out), n_pix);
    else if (dtype == MLA_F32)
        hipLaunchKernelGGL((conv1_kernel<bf16_t, float>), g, b, 0, s, static_cast<const bf16_t*>(x), w, bias, static_cast<float*>(out), n_pix);
    else
        hipLaunchKernelGGL((conv1_kernel<bf16_t, bf16_t>), g, b, 0, s, static_cast<const bf16_t*>(x), w, bias, static_cast<bf16_t*>(out), n_pix);
    MLA_LAUNCH_OK("conv1_kernel");
    return MLA_OK;
}

extern "C" int mla_vggish_conv(int layer, const void* in, const void* w_repacked, const float* bias, void* out,
                               int64_t n, int dtype, mla_stream_t stream) {
    MLA_REQUIRE(n >= 0, MLA_E_ARG, "n %lld", (long long)n);
    if (n == 0) return MLA_OK;
    MLA_REQUIRE(in && w_repacked && bias && out, MLA_E_ARG, "null conv buffers");
    MLA_REQUIRE(mla::aligned(in, 16) && mla::aligned(w_repacked, 16) && mla::aligned(out, 4), MLA_E_ARG, "conv buffers must be 16-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == MLA_F32) return conv_layer<float>(layer, in, w_repacked, bias, out, n, s);
    if (dtype == MLA_BF16) return conv_layer<bf16_t>(layer, in, w_repacked, bias, out, n, s);
    if (dtype == MLA_BF16X3) return conv_layer_split(layer, in, w_repacked, bias, out, n, s);
    return mla::fail(MLA_E_DTYPE, "conv dtype %d", dtype);
}
