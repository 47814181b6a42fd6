// gemm.hip -- out[M, N] = act(A[M, K] . W[N, K]^T + bias[N]) on the matrix cores, for every
// torch.nn.Linear on the hot path: VGG.embeddings (vggish.py:13-19: 12288->4096->4096->128,
// each followed by ReLU) and the MLA head's fc / fcv layers (model.py:207-210, :230, :255).
// Both operands are K-contiguous exactly as PyTorch stores them (activations row-major,
// Linear.weight = [out, in]), so no transposition is ever materialised.
//
// Same tile machinery as conv.hip (mma_core.h): 512 threads = 2 (M) x 4 (N) waves, tile
// 128 x (64 NS), 128-byte K chunks (64 bf16 / 32 f32) double-buffered in LDS with the next
// chunk's global loads in flight behind the MFMAs; M / N / K tails are zero-filled on load and
// masked on store. bf16 uses v_mfma_f32_16x16x32_bf16, f32 the exact v_mfma_f32_16x16x4_f32.
#include <type_traits>

#include "common.h"
#include "mma_core.h"

namespace {

using namespace mma;

constexpr int kThreads = 512;

// Diagnostic build only (-DMLA_GEMM_STAMPS=1): s_memtime stamps of stage 100 of waves 0 and 4 of the first 8 workgroups (see conv.hip)
#ifndef MLA_GEMM_STAMPS
#define MLA_GEMM_STAMPS 0
#endif
#if MLA_GEMM_STAMPS
__device__ unsigned long long g_gemm_stamps[8][2][8];
#define MLA_GSTAMP(k) do { if (stamp_on) stamps[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MLA_GSTAMP(k) do { } while (0)
#endif
#ifndef MLA_GEMM_PRIO
#define MLA_GEMM_PRIO 3             // 3: burst priorities (see conv.hip), 1: waves 4-7 static s_setprio 1
#endif
#ifndef MLA_GEMM_STAGGER
#define MLA_GEMM_STAGGER 1          // 0: A/B builds without the half-stage stagger of waves 4-7 (scripts/build_variant.py)
#endif

// DMA: both operand tiles go global -> LDS by LDS-DMA (buffer_load ... lds, 1 KiB = 8 tile rows per wave-instruction, the
// tile's XOR swizzle applied on the source side) instead of through registers + ds_write: no staging registers, no
// write pass in front of the barrier. Needs K % (64 or 32) == 0: an out-of-range DMA lane is dropped, not zero-filled
// (rows past M / N only feed masked outputs; a K tail would feed valid ones).
#ifndef MLA_GEMM_TILE320
#define MLA_GEMM_TILE320 1
#endif
#ifndef MLA_GEMM_RING
#define MLA_GEMM_RING 1             // 0: A/B builds without the four-stage ring form of the small-batch 128 x 128 tiles
#endif
// timing experiment only (wrong results): -DMLA_GEMM_KWRAP=8 keeps every tile's operands inside 8 K stages, i.e. L2-resident
#ifdef MLA_GEMM_KWRAP
#define MLA_GEMM_KW(s) ((s) & (MLA_GEMM_KWRAP - 1))
#else
#define MLA_GEMM_KW(s) (s)
#endif
template <typename T, typename TO, int MS, int NS, bool RELU, bool DMA = false>
__global__ __launch_bounds__(kThreads, 2) void gemm_kernel(const T* __restrict__ A, int64_t lda,
                                                            const T* __restrict__ W, int64_t ldw,
                                                            const float* __restrict__ bias, TO* __restrict__ out,
                                                            int64_t ldo, int M, int N, int K, float* __restrict__ partial,
                                                            int seg, int osplit) {
    constexpr int PER = Elem<T>::kPerChunk, KC = Elem<T>::kPerRow;
    constexpr int kMS = MS, kBM = 2 * MS * 16;                  // 2 (M) x 4 (N) waves: tile (32 MS) x (64 NS)
    constexpr int BN = 4 * NS * 16;
    constexpr int A_BYTES = kBM * kRowBytes, B_BYTES = BN * kRowBytes;
    constexpr int AP = kBM * 8 / kThreads;                      // 16-byte A pieces per thread and stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;
    char* sB = smem + 2 * A_BYTES;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 2, wn = wave & 3;
    const int r = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.x * kBM, n0 = blockIdx.y * BN;
    const int abase = tile_off(wm * kMS * 16 + r, q);
    const int bbase = tile_off(wn * NS * 16 + r, q);

    f32x4 acc[kMS][NS];
    _Pragma("unroll") for (int i = 0; i < kMS; ++i)
        _Pragma("unroll") for (int j = 0; j < NS; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr uint32_t ESZ = sizeof(T);
    const int a_rows = M - m0 < kBM ? M - m0 : kBM, w_rows = N - n0 < BN ? N - n0 : BN;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(A) + size_t(m0) * lda, 0,
                                                                          uint32_t(a_rows) * uint32_t(lda) * ESZ, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(W) + size_t(n0) * ldw, 0,
                                                                          uint32_t(w_rows) * uint32_t(ldw) * ESZ, 0x00020000);
    u32x4 areg[AP], breg[NS];
    // seg > 0 ("bf16x3", see conv.hip): W holds [w_hi | w_lo | w_hi] per segment of `seg` K-chunks and A [a_hi | a_lo];
    // weight stage s = 3 seg b + r pairs with activation stage 2 seg b + (r < seg ? r : r - seg). K counts W columns.
    // register-staged path (K tails, 32-bit descriptor overflow): loads through buffer descriptors whose range check returns zeros
    // for rows past M / N and columns past K -- no lane-dependent branches around the loads (hipcc would wait for each one)
    auto gload = [&](int s) {
        int sa = s;
        if (seg) {
            const int blk = s / (3 * seg), rr = s - blk * 3 * seg;
            sa = blk * 2 * seg + (rr < seg ? rr : rr - seg);
        }
        _Pragma("unroll") for (int p = 0; p < AP; ++p) {
            const int piece = t + kThreads * p, row = piece >> 3, ch = piece & 7;
            const int k = sa * KC + ch * PER;
            const bool ok = m0 + row < M && s * KC + ch * PER < K;
            areg[p] = __builtin_amdgcn_raw_buffer_load_b128(ra, ok ? int((uint32_t(row) * uint32_t(lda) + k) * sizeof(T)) : int(0x7fffff00), 0, 0);
        }
        _Pragma("unroll") for (int p = 0; p < NS; ++p) {
            const int piece = t + kThreads * p, n = piece >> 3, ch = piece & 7;
            const int k = s * KC + ch * PER;
            const bool ok = n0 + n < N && k < K;
            breg[p] = __builtin_amdgcn_raw_buffer_load_b128(rw, ok ? int((uint32_t(n) * uint32_t(ldw) + k) * sizeof(T)) : int(0x7fffff00), 0, 0);
        }
    };
    auto lwrite = [&](int buf) {
        _Pragma("unroll") for (int p = 0; p < AP; ++p) {
            const int piece = t + kThreads * p;
            lds_write16(sA, buf * A_BYTES + tile_off(piece >> 3, piece & 7), areg[p]);
        }
        _Pragma("unroll") for (int p = 0; p < NS; ++p) {
            const int piece = t + kThreads * p;
            lds_write16(sB, buf * B_BYTES + tile_off(piece >> 3, piece & 7), breg[p]);
        }
    };

    auto dma = [&](int s, int buf) {
        int sa = s;
        if (seg) {
            const int blk = s / (3 * seg), rr = s - blk * 3 * seg;
            sa = blk * 2 * seg + (rr < seg ? rr : rr - seg);
        }
        const int slot = lane & 7;
        _Pragma("unroll") for (int p = 0; p < kBM / 64; ++p) {
            const int row = 8 * (wave + 8 * p) + (lane >> 3);
            const int chunk = slot ^ (((row >> 1) & 3) << 1);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(sA + buf * A_BYTES + 8 * (wave + 8 * p) * kRowBytes),
                                                     16, int((uint32_t(row) * uint32_t(lda) + chunk * PER) * ESZ), int(MLA_GEMM_KW(sa) * KC * ESZ), 0, 0);
        }
        _Pragma("unroll") for (int p = 0; p < BN / 64; ++p) {
            const int row = 8 * (wave + 8 * p) + (lane >> 3);
            const int chunk = slot ^ (((row >> 1) & 3) << 1);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(sB + buf * B_BYTES + 8 * (wave + 8 * p) * kRowBytes),
                                                     16, int((uint32_t(row) * uint32_t(ldw) + chunk * PER) * ESZ), int(MLA_GEMM_KW(s) * KC * ESZ), 0, 0);
        }
    };

    // split-K: blockIdx.z owns a contiguous range of K stages and writes raw partial sums
    const int all_stages = (K + KC - 1) / KC;
    const int per_split = (all_stages + int(gridDim.z) - 1) / int(gridDim.z);
    const int s_begin = int(blockIdx.z) * per_split;
    const int stages = (s_begin + per_split < all_stages ? s_begin + per_split : all_stages);
    if (DMA) {
        dma(s_begin, s_begin & 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // DMA data is ordered for ds_read by vmcnt + barrier only
    } else {
        gload(s_begin);
        lwrite(s_begin & 1);
    }
    __syncthreads();
    // Stagger (DMA path: one workgroup per CU): waves 4-7, the SIMD partners of waves 0-3, run half a stage behind -- their
    // k-step-1 fragments are read before the barrier that ends a stage and multiplied after it -- and at s_setprio 1; see
    // conv.hip (Cfg::STAGGER). Same products in the same order per accumulator: bit-identical results.
    constexpr bool STAGGER = DMA && MLA_GEMM_STAGGER;
    const bool late = STAGGER && __builtin_amdgcn_readfirstlane(wave) >= 4;
    if (STAGGER && late) __builtin_amdgcn_s_setprio(1);
#if MLA_GEMM_STAMPS
    unsigned long long stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    auto run = [&](auto late_c) {
        constexpr bool LATE = decltype(late_c)::value;
        u32x4 af[kMS], bf[NS];
        for (int s = s_begin; s < stages; ++s) {
            const int buf = s & 1;
#if MLA_GEMM_STAMPS
            const bool stamp_on = s == 100 && (wave == 0 || wave == 4) && blockIdx.x < 8 && blockIdx.y == 0 && kMS == 8;
#endif
            auto stage_next = [&]() {              // next stage's operands; issued after the wave's first reads / carried-over burst
                if (s + 1 < stages) {
                    if (DMA) dma(s + 1, buf ^ 1);  // the other buffer's last readers passed the barrier that ended stage s - 1
                    else gload(s + 1);
                }
            };
            auto rd = [&](int ks) {
                _Pragma("unroll") for (int i = 0; i < kMS; ++i)
                    af[i] = lds_read16(sA, buf * A_BYTES + (abase ^ (ks << 6)) + i * 16 * kRowBytes);
                _Pragma("unroll") for (int j = 0; j < NS; ++j)
                    bf[j] = lds_read16(sB, buf * B_BYTES + (bbase ^ (ks << 6)) + j * 16 * kRowBytes);
            };
            auto mm = [&]() {                      // coarse phases: all reads, then one MFMA burst (see conv.hip)
                __builtin_amdgcn_sched_barrier(0);
                _Pragma("unroll") for (int i = 0; i < kMS; ++i)
                    _Pragma("unroll") for (int j = 0; j < NS; ++j) mma_step<T>(af[i], bf[j], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
            };
            // burst priorities (conv.hip, MLA_CONV_PRIO 3): carried-over burst 3 > early k-step 0 2 > late k-step 0 1 > early k-step 1 0,
            // so that no MFMA burst is preempted by the SIMD partner and the bursts alternate late, early, late, early
            if constexpr (!LATE) {
                if (STAGGER && MLA_GEMM_PRIO == 3) __builtin_amdgcn_s_setprio(2);
                MLA_GSTAMP(0);
                rd(0);
                MLA_GSTAMP(1);
                stage_next();
                MLA_GSTAMP(2);
                mm();
                MLA_GSTAMP(3);
                if (STAGGER && MLA_GEMM_PRIO == 3) __builtin_amdgcn_s_setprio(0);
                rd(1);
                MLA_GSTAMP(4);
                mm();
                MLA_GSTAMP(5);
            } else {
                if (MLA_GEMM_PRIO == 3) __builtin_amdgcn_s_setprio(3);
                MLA_GSTAMP(0);
                if (s > s_begin) mm();             // k-step 1 of the previous stage
                MLA_GSTAMP(1);
                stage_next();
                MLA_GSTAMP(2);
                if (MLA_GEMM_PRIO == 3) __builtin_amdgcn_s_setprio(1);
                rd(0); mm();
                MLA_GSTAMP(3);
                rd(1);
                MLA_GSTAMP(4);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the reads have left the buffer before the next DMA may land in it
                MLA_GSTAMP(5);
            }
            if (!DMA && s + 1 < stages) lwrite(buf ^ 1);   // the other buffer was last read before the previous barrier
            if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            MLA_GSTAMP(6);
            __syncthreads();
            MLA_GSTAMP(7);
        }
        if constexpr (LATE) {
            if (stages > s_begin) {
                __builtin_amdgcn_sched_barrier(0);
                _Pragma("unroll") for (int i = 0; i < kMS; ++i)
                    _Pragma("unroll") for (int j = 0; j < NS; ++j) mma_step<T>(af[i], bf[j], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    if constexpr (STAGGER) {
        if (late) run(std::true_type{});
        else run(std::false_type{});
    } else {
        run(std::false_type{});
    }

#if MLA_GEMM_STAMPS
    if ((wave == 0 || wave == 4) && lane == 0 && blockIdx.x < 8 && blockIdx.y == 0 && kMS == 8 && stages > 100)
        for (int k = 0; k < 8; ++k) g_gemm_stamps[blockIdx.x][wave >> 2][k] = stamps[k];
#endif
    if (partial) {                          // raw sums of this K range; bias / activation happen in the reduction
        float* pout = partial + size_t(blockIdx.z) * M * N;
        _Pragma("unroll") for (int j = 0; j < NS; ++j) {
            const int n = n0 + (wn * NS + j) * 16 + r;
            if (n >= N) continue;
            _Pragma("unroll") for (int i = 0; i < kMS; ++i) {
                const float v[4] = {acc[i][j].x, acc[i][j].y, acc[i][j].z, acc[i][j].w};
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {
                    const int m = m0 + (wm * kMS + i) * 16 + 4 * q + e;
                    if (m < M) pout[size_t(m) * N + n] = v[e];
                }
            }
        }
        return;
    }
    _Pragma("unroll") for (int j = 0; j < NS; ++j) {
        const int n = n0 + (wn * NS + j) * 16 + r;
        if (n >= N) continue;
        const float b = bias ? bias[n] : 0.f;
        _Pragma("unroll") for (int i = 0; i < kMS; ++i) {
            const float v[4] = {acc[i][j].x, acc[i][j].y, acc[i][j].z, acc[i][j].w};
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {
                const int m = m0 + (wm * kMS + i) * 16 + 4 * q + e;
                if (m < M) {
                    float y = v[e] + b;
                    if (RELU) y = fmaxf(y, 0.f);
                    if (sizeof(TO) == 2 && osplit) {              // [hi(N) | lo(N)] row of 2 N bf16
                        const float hi = bf2f(f2bf(y));
                        store_elem<TO>(out + size_t(m) * ldo + n, hi);
                        store_elem<TO>(out + size_t(m) * ldo + N + n, y - hi);
                    } else {
                        store_elem<TO>(out + size_t(m) * ldo + n, y);
                    }
                }
            }
        }
    }
}

// ---- small-batch form: 128 x 128 tiles behind a FOUR-stage LDS ring --------------------------------------------------------
// At ~1 000 rows (BASELINE config 3 read literally: 1 020 clips per step) the FC layers have only enough 128 x 128 tiles to
// give every CU one, each tile streams its whole K range (32 KB per 64-deep stage for 2.1 MFLOP: the MFMA work of a stage is a
// fifth of its fill time), and the double-buffered kernel above keeps ONE stage in flight per CU: the layer then runs at
// L2-latency x 32 KB (measured 30 GB/s per CU; FC1 208 us for 1 020 rows, 2.7x its per-row time at 10 240 rows). This kernel keeps
// THREE stages (96 KB) in flight: stage s + 3 is requested as soon as the barrier that opens stage s has shown that everybody
// left stage s - 1's buffer; each wave waits for its own pieces of stage s with a counted vmcnt (8 / 4 / 0 younger pieces may
// stay outstanding) and the barrier publishes the other waves' pieces. One barrier per stage. Products per accumulator in the
// same order as every other tile configuration (stage by stage, k-step 0 then 1), so a row's result does not depend on which
// kernel -- i.e. which batch size -- computed it (bit-identical, tested).
// Workgroup -> tile: the 8 XCDs each take a band of column tiles for ALL row tiles (block ids b, b + 8, ... share an XCD), so
// a weight tile is fetched into ONE L2 and the (small) activation matrix into all of them, instead of the other way round.
template <typename T, typename TO, bool RELU>
__global__ __launch_bounds__(kThreads, 2) void gemm_ring_kernel(const T* __restrict__ A, int64_t lda, const T* __restrict__ W,
                                                                 int64_t ldw, const float* __restrict__ bias, TO* __restrict__ out,
                                                                 int64_t ldo, int M, int N, int K, float* __restrict__ partial, int seg,
                                                                 int osplit) {
    constexpr int PER = Elem<T>::kPerChunk, KC = Elem<T>::kPerRow;
    constexpr int kMS = 4, NS = 2, kBM = 128, BN = 128, RING = 4;
    constexpr int A_BYTES = kBM * kRowBytes, B_BYTES = BN * kRowBytes, STAGE_BYTES = A_BYTES + B_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 2, wn = wave & 3;
    const int r = lane & 15, q = lane >> 4;
    int bx = blockIdx.x, by = blockIdx.y;
    if (gridDim.y % 8 == 0) {                              // XCD k (block ids = k mod 8) <- column tiles [k n/8, (k + 1) n/8) x all row tiles
        const int id = blockIdx.x + gridDim.x * blockIdx.y, per = gridDim.y / 8, j = id >> 3;
        by = (id & 7) * per + j % per;
        bx = j / per;
    }
    const int m0 = bx * kBM, n0 = by * BN;
    const int abase = tile_off(wm * kMS * 16 + r, q);
    const int bbase = tile_off(wn * NS * 16 + r, q);

    f32x4 acc[kMS][NS];
    _Pragma("unroll") for (int i = 0; i < kMS; ++i)
        _Pragma("unroll") for (int j = 0; j < NS; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr uint32_t ESZ = sizeof(T);
    const int a_rows = M - m0 < kBM ? M - m0 : kBM, w_rows = N - n0 < BN ? N - n0 : BN;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(A) + size_t(m0) * lda, 0,
                                                                          uint32_t(a_rows) * uint32_t(lda) * ESZ, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(W) + size_t(n0) * ldw, 0,
                                                                          uint32_t(w_rows) * uint32_t(ldw) * ESZ, 0x00020000);
    // four 1 KiB pieces per wave and stage (2 of A, 2 of W): rows 8 (wave + 8 p) ..+7, source-side swizzle as in gemm_kernel
    auto dma = [&](int s, int buf) {
        int sa = s;
        if (seg) {
            const int blk = s / (3 * seg), rr = s - blk * 3 * seg;
            sa = blk * 2 * seg + (rr < seg ? rr : rr - seg);
        }
        char* sA = smem + buf * STAGE_BYTES;
        char* sB = sA + A_BYTES;
        const int slot = lane & 7;
        _Pragma("unroll") for (int p = 0; p < 2; ++p) {
            const int row = 8 * (wave + 8 * p) + (lane >> 3);
            const int chunk = slot ^ (((row >> 1) & 3) << 1);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(sA + 8 * (wave + 8 * p) * kRowBytes), 16,
                                                     int((uint32_t(row) * uint32_t(lda) + chunk * PER) * ESZ), int(sa * KC * ESZ), 0, 0);
        }
        _Pragma("unroll") for (int p = 0; p < 2; ++p) {
            const int row = 8 * (wave + 8 * p) + (lane >> 3);
            const int chunk = slot ^ (((row >> 1) & 3) << 1);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(sB + 8 * (wave + 8 * p) * kRowBytes), 16,
                                                     int((uint32_t(row) * uint32_t(ldw) + chunk * PER) * ESZ), int(s * KC * ESZ), 0, 0);
        }
    };

    const int all_stages = K / KC;                          // K % KC == 0 (checked by the launcher)
    const int per_split = (all_stages + int(gridDim.z) - 1) / int(gridDim.z);
    const int s_begin = int(blockIdx.z) * per_split;
    const int s_end = s_begin + per_split < all_stages ? s_begin + per_split : all_stages;
    _Pragma("unroll") for (int p = 0; p < RING - 1; ++p)
        if (s_begin + p < s_end) dma(s_begin + p, p);
    for (int s = s_begin; s < s_end; ++s) {
        const int left = s_end - 1 - s;                     // stages requested after s that may still be in flight (each 4 pieces of this wave)
        if (left >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (left == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                    // stage s is complete in LDS; nobody reads stage s - 1's buffer any more
        if (s + RING - 1 < s_end) dma(s + RING - 1, (s - s_begin + RING - 1) % RING);
        const char* sA = smem + ((s - s_begin) % RING) * STAGE_BYTES;
        const char* sB = sA + A_BYTES;
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {
            u32x4 af[kMS], bf[NS];
            _Pragma("unroll") for (int i = 0; i < kMS; ++i) af[i] = lds_read16(sA, (abase ^ (ks << 6)) + i * 16 * kRowBytes);
            _Pragma("unroll") for (int j = 0; j < NS; ++j) bf[j] = lds_read16(sB, (bbase ^ (ks << 6)) + j * 16 * kRowBytes);
            _Pragma("unroll") for (int i = 0; i < kMS; ++i)
                _Pragma("unroll") for (int j = 0; j < NS; ++j) mma_step<T>(af[i], bf[j], acc[i][j]);
        }
    }

    if (partial) {                          // raw sums of this K range; bias / activation happen in the reduction
        float* pout = partial + size_t(blockIdx.z) * M * N;
        _Pragma("unroll") for (int j = 0; j < NS; ++j) {
            const int n = n0 + (wn * NS + j) * 16 + r;
            if (n >= N) continue;
            _Pragma("unroll") for (int i = 0; i < kMS; ++i) {
                const float v[4] = {acc[i][j].x, acc[i][j].y, acc[i][j].z, acc[i][j].w};
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {
                    const int m = m0 + (wm * kMS + i) * 16 + 4 * q + e;
                    if (m < M) pout[size_t(m) * N + n] = v[e];
                }
            }
        }
        return;
    }
    _Pragma("unroll") for (int j = 0; j < NS; ++j) {
        const int n = n0 + (wn * NS + j) * 16 + r;
        if (n >= N) continue;
        const float b = bias ? bias[n] : 0.f;
        _Pragma("unroll") for (int i = 0; i < kMS; ++i) {
            const float v[4] = {acc[i][j].x, acc[i][j].y, acc[i][j].z, acc[i][j].w};
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {
                const int m = m0 + (wm * kMS + i) * 16 + 4 * q + e;
                if (m < M) {
                    float y = v[e] + b;
                    if (RELU) y = fmaxf(y, 0.f);
                    if (sizeof(TO) == 2 && osplit) {              // [hi(N) | lo(N)] row of 2 N bf16
                        const float hi = bf2f(f2bf(y));
                        store_elem<TO>(out + size_t(m) * ldo + n, hi);
                        store_elem<TO>(out + size_t(m) * ldo + N + n, y - hi);
                    } else {
                        store_elem<TO>(out + size_t(m) * ldo + n, y);
                    }
                }
            }
        }
    }
}

// out = act(sum over splits + bias), fixed order
template <typename TO>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, int splits, int64_t M, int N,
                                                            const float* __restrict__ bias, int relu, TO* __restrict__ out, int64_t ldo) {
    const int64_t total = M * N;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < total; i += int64_t(gridDim.x) * 256) {
        const int64_t m = i / N;
        const int n = int(i - m * N);
        float v = bias ? bias[n] : 0.f;
        for (int k = 0; k < splits; ++k) v += partial[size_t(k) * total + i];
        if (relu) v = fmaxf(v, 0.f);
        store_elem<TO>(out + m * ldo + n, v);
    }
}

template <typename T, typename TO, int MS, int NS, bool RELU>
int launch(const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias, void* out, int64_t ldo,
           int64_t M, int64_t N, int64_t K, hipStream_t s, int splits = 1, float* partial = nullptr, int seg = 0, int osplit = 0) {
    constexpr int kBM = 2 * MS * 16, BN = 4 * NS * 16;
    constexpr int lds = 2 * (kBM + BN) * kRowBytes;
    // LDS-DMA staging when no K tail needs zero filling and the descriptors' 32-bit byte ranges suffice
    constexpr int KCE = mma::Elem<T>::kPerRow;
    MLA_REQUIRE(uint64_t(kBM) * uint64_t(lda) * sizeof(T) < (1ull << 31) && uint64_t(BN) * uint64_t(ldw) * sizeof(T) < (1ull << 31), MLA_E_SHAPE,
                "GEMM row pitch too large for 32-bit buffer offsets (lda %lld, ldw %lld)", (long long)lda, (long long)ldw);
    const bool use_dma = K % KCE == 0;
    auto kern = use_dma ? gemm_kernel<T, TO, MS, NS, RELU, true> : gemm_kernel<T, TO, MS, NS, RELU, false>;
    MLA_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const dim3 grid{unsigned((M + kBM - 1) / kBM), unsigned((N + BN - 1) / BN), unsigned(splits)};
    hipLaunchKernelGGL(kern, grid, dim3(kThreads), lds, s, static_cast<const T*>(a), lda, static_cast<const T*>(w), ldw,
                       bias, static_cast<TO*>(out), ldo, int(M), int(N), int(K), splits > 1 ? partial : nullptr, seg, osplit);
    MLA_LAUNCH_OK("gemm_kernel");
    if (splits > 1) {
        const int64_t total = M * N;
        hipLaunchKernelGGL(splitk_reduce_kernel<TO>, dim3(unsigned((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)), dim3(256),
                           0, s, partial, splits, M, int(N), bias, int(RELU), static_cast<TO*>(out), ldo);
        MLA_LAUNCH_OK("splitk_reduce_kernel");
    }
    return MLA_OK;
}

template <typename T, typename TO, bool RELU>
int launch_ring(const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias, void* out, int64_t ldo,
                int64_t M, int64_t N, int64_t K, hipStream_t s, int splits = 1, float* partial = nullptr, int seg = 0, int osplit = 0) {
    constexpr int lds = 4 * (128 + 128) * kRowBytes;                      // four stages of a 128 x 128 tile: 128 KB
    MLA_REQUIRE(uint64_t(128) * uint64_t(lda) * sizeof(T) < (1ull << 31) && uint64_t(128) * uint64_t(ldw) * sizeof(T) < (1ull << 31), MLA_E_SHAPE,
                "GEMM row pitch too large for 32-bit buffer offsets (lda %lld, ldw %lld)", (long long)lda, (long long)ldw);
    MLA_REQUIRE(K % mma::Elem<T>::kPerRow == 0, MLA_E_SHAPE, "the ring GEMM stages whole %d-element K chunks", mma::Elem<T>::kPerRow);
    auto kern = gemm_ring_kernel<T, TO, RELU>;
    MLA_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const dim3 grid{unsigned((M + 127) / 128), unsigned((N + 127) / 128), unsigned(splits)};
    hipLaunchKernelGGL(kern, grid, dim3(kThreads), lds, s, static_cast<const T*>(a), lda, static_cast<const T*>(w), ldw, bias,
                       static_cast<TO*>(out), ldo, int(M), int(N), int(K), splits > 1 ? partial : nullptr, seg, osplit);
    MLA_LAUNCH_OK("gemm_ring_kernel");
    if (splits > 1) {
        const int64_t total = M * N;
        hipLaunchKernelGGL(splitk_reduce_kernel<TO>, dim3(unsigned((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)), dim3(256),
                           0, s, partial, splits, M, int(N), bias, int(RELU), static_cast<TO*>(out), ldo);
        MLA_LAUNCH_OK("splitk_reduce_kernel");
    }
    return MLA_OK;
}

template <typename T, typename TO>
int dispatch(const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias, void* out, int64_t ldo,
             int64_t M, int64_t N, int64_t K, bool relu, hipStream_t s, int seg = 0, int osplit = 0) {
    const bool wide = N > 128 && (N % 256 == 0 || N % 256 > 128);     // 256-wide tiles unless they waste > half a tile
    const bool tall = wide && M >= 4096 && N >= 1024;                  // 256 x 256 tiles once they still fill the chip
#if MLA_GEMM_TILE320
    if (tall && K % mma::Elem<T>::kPerRow == 0) {      // LDS-DMA path only (the register-staged form of this tile spills)
        // 320 x 256 tiles (waves 2 x 4, 10 x 4 accumulator tiles each): 10 240 rows x 4096 columns are 512 tiles = exactly two
        // rounds on 256 CUs (256 x 256: 640 tiles = 2.5 rounds), and the tile moves 72 KB per K stage for 1.25x the FLOPs of the
        // 64 KB one -- the GEMM is bound by the CU's L2 -> LDS fill rate (profiles/r02_conv_stamps.txt), so both count. Taken when
        // its rounds x bytes per stage beat the 256-row tiling's (K order per output element unchanged: bit-identical).
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const int64_t n_tiles = (N + 255) / 256;
        const int64_t t256 = ((M + 255) / 256) * n_tiles, t320 = ((M + 319) / 320) * n_tiles;
        const int64_t full = t256 / cus, rest = t256 - full * cus;
        const double cost256 = 64.0 * double(full) + (rest == 0 ? 0.0 : (full >= 1 && 2 * rest <= cus && (full * cus) % n_tiles == 0 ? 48.0 : 64.0));
        const double cost320 = 72.0 * double((t320 + cus - 1) / cus);
        if (cost320 < cost256)
            return relu ? launch<T, TO, 10, 4, true>(a, lda, w, ldw, bias, out, ldo, M, N, K, s, 1, nullptr, seg, osplit)
                        : launch<T, TO, 10, 4, false>(a, lda, w, ldw, bias, out, ldo, M, N, K, s, 1, nullptr, seg, osplit);
    }
#endif
    if (tall) {
        // Tile quantisation: 256 x 256 tiles run one per CU, so e.g. 640 tiles on 256 CUs take 3 rounds for 2.5 rounds of
        // work. When the last round would be at most half full, the rows of the whole rounds keep the tall tiles and the
        // remaining rows run as 128-row tiles (twice as many, same CUs busy): 2 + 0.5 rounds instead of 3.
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const int64_t n_tiles = (N + 255) / 256, m_tiles = (M + 255) / 256;
        const int64_t full_rounds = m_tiles * n_tiles / cus, rest = m_tiles * n_tiles - full_rounds * cus;
        int64_t m_tall = M;
        if (full_rounds >= 1 && rest > 0 && 2 * rest <= cus && (full_rounds * cus) % n_tiles == 0) m_tall = full_rounds * cus / n_tiles * 256;
        int rc = relu ? launch<T, TO, 8, 4, true>(a, lda, w, ldw, bias, out, ldo, m_tall, N, K, s, 1, nullptr, seg, osplit)
                      : launch<T, TO, 8, 4, false>(a, lda, w, ldw, bias, out, ldo, m_tall, N, K, s, 1, nullptr, seg, osplit);
        if (rc != MLA_OK || m_tall == M) return rc;
        const T* a2 = static_cast<const T*>(a) + m_tall * lda;
        TO* out2 = static_cast<TO*>(out) + m_tall * ldo;
        return relu ? launch<T, TO, 4, 4, true>(a2, lda, w, ldw, bias, out2, ldo, M - m_tall, N, K, s, 1, nullptr, seg, osplit)
                    : launch<T, TO, 4, 4, false>(a2, lda, w, ldw, bias, out2, ldo, M - m_tall, N, K, s, 1, nullptr, seg, osplit);
    }
    // 128 x 256 tiles that would leave half the CUs idle (small batches: 1 020 rows x 4096 columns = 128 tiles) run as
    // 128 x 128 tiles instead
    if (wide) {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (((M + 127) / 128) * ((N + 255) / 256) * 2 <= cus) {
#if MLA_GEMM_RING
            if (K % mma::Elem<T>::kPerRow == 0 && K / mma::Elem<T>::kPerRow >= 8)       // long K streamed by few tiles: keep three stages in flight
                return relu ? launch_ring<T, TO, true>(a, lda, w, ldw, bias, out, ldo, M, N, K, s, 1, nullptr, seg, osplit)
                            : launch_ring<T, TO, false>(a, lda, w, ldw, bias, out, ldo, M, N, K, s, 1, nullptr, seg, osplit);
#endif
            return relu ? launch<T, TO, 4, 2, true>(a, lda, w, ldw, bias, out, ldo, M, N, K, s, 1, nullptr, seg, osplit)
                        : launch<T, TO, 4, 2, false>(a, lda, w, ldw, bias, out, ldo, M, N, K, s, 1, nullptr, seg, osplit);
        }
    }
    if (wide) return relu ? launch<T, TO, 4, 4, true>(a, lda, w, ldw, bias, out, ldo, M, N, K, s, 1, nullptr, seg, osplit)
                          : launch<T, TO, 4, 4, false>(a, lda, w, ldw, bias, out, ldo, M, N, K, s, 1, nullptr, seg, osplit);
    // narrow layers (N <= 128: the embeddings' last Linear 4096 -> 128, the head's 128 -> 600 tail tiles) at batch sizes where
    // 128-row tiles leave most CUs idle: 64-row tiles double the workgroups (the K loop per output element is unchanged, so the
    // bits are too)
    {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (((M + 127) / 128) * ((N + 127) / 128) < cus && M > 64)
            return relu ? launch<T, TO, 2, 2, true>(a, lda, w, ldw, bias, out, ldo, M, N, K, s, 1, nullptr, seg, osplit)
                        : launch<T, TO, 2, 2, false>(a, lda, w, ldw, bias, out, ldo, M, N, K, s, 1, nullptr, seg, osplit);
    }
    return relu ? launch<T, TO, 4, 2, true>(a, lda, w, ldw, bias, out, ldo, M, N, K, s, 1, nullptr, seg, osplit)
                : launch<T, TO, 4, 2, false>(a, lda, w, ldw, bias, out, ldo, M, N, K, s, 1, nullptr, seg, osplit);
}

}  // namespace

#if MLA_GEMM_STAMPS
extern "C" int mla_debug_gemm_stamps(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_gemm_stamps), sizeof(g_gemm_stamps)) == hipSuccess ? 0 : -4;
}
#endif

// Split-K form for reductions over a long K with few output tiles (weight gradients dW = dZ^T . X:
// M, N = layer widths, K = batch rows): `splits` K ranges accumulate into workspace[splits][M][N]
// and are summed in fixed order (deterministic). f32 only.
extern "C" int mla_linear_splitk(const float* a, int64_t lda, const float* w, int64_t ldw, const float* bias, float* out,
                                 int64_t ldo, int64_t M, int64_t N, int64_t K, int relu, int splits, float* workspace,
                                 int64_t workspace_floats, mla_stream_t stream) {
    MLA_REQUIRE(a && w && out && workspace && M > 0 && N > 0 && K > 0 && splits >= 1 && splits <= 64, MLA_E_ARG, "bad split-K GEMM arguments");
    MLA_REQUIRE(K % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0 && lda >= K && ldw >= K && ldo >= N, MLA_E_SHAPE, "split-K GEMM: 16-byte rows required");
    MLA_REQUIRE(mla::aligned(a, 16) && mla::aligned(w, 16), MLA_E_ARG, "GEMM operands must be 16-byte aligned");
    MLA_REQUIRE(workspace_floats >= int64_t(splits) * M * N, MLA_E_ARG, "split-K workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    return relu ? launch<float, float, 4, 2, true>(a, lda, w, ldw, bias, out, ldo, M, N, K, s, splits, workspace)
                : launch<float, float, 4, 2, false>(a, lda, w, ldw, bias, out, ldo, M, N, K, s, splits, workspace);
}

// Fixed K split for NARROW forward layers (N <= 128 with a long K: VGGish's last Linear 4096 -> 128): with one column tile there
// are M / 128 workgroups for 256 CUs -- 80 at 10 240 rows, 16 at 1 020 -- each walking all of K. `splits` K ranges (the CALLER
// fixes the number per layer, never per batch) run as separate workgroups; their float32 partial
// sums are added in range order, then bias and activation. The order of every addition is a function of (K, splits) alone, so
// a row's result is the same in every batch (what split-K chosen by batch size would break).
extern "C" int mla_linear_ksplit(const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias, void* out, int64_t ldo,
                                 int64_t M, int64_t N, int64_t K, int dtype, int out_dtype, int relu, int splits, float* workspace,
                                 int64_t workspace_floats, mla_stream_t stream) {
    MLA_REQUIRE(M >= 0 && N > 0 && K > 0 && splits >= 1 && splits <= 64, MLA_E_ARG, "bad K-split GEMM arguments");
    if (M == 0) return MLA_OK;
    MLA_REQUIRE(a && w && out && workspace, MLA_E_ARG, "null GEMM operand");
    MLA_REQUIRE(dtype == MLA_F32 || dtype == MLA_BF16, MLA_E_DTYPE, "GEMM dtype %d", dtype);
    MLA_REQUIRE(out_dtype == MLA_F32 || (out_dtype == MLA_BF16 && dtype == MLA_BF16), MLA_E_DTYPE, "GEMM out dtype %d for compute dtype %d", out_dtype, dtype);
    const int kc = dtype == MLA_F32 ? 32 : 64;
    MLA_REQUIRE(K % (int64_t(kc) * splits) == 0 && lda % (kc / 8) == 0 && ldw % (kc / 8) == 0 && lda >= K && ldw >= K && ldo >= N, MLA_E_SHAPE,
                "K-split GEMM: K %lld must be a multiple of %d x splits %d (whole stages per range)", (long long)K, kc, splits);
    MLA_REQUIRE(mla::aligned(a, 16) && mla::aligned(w, 16), MLA_E_ARG, "GEMM operands must be 16-byte aligned");
    MLA_REQUIRE(workspace_floats >= int64_t(splits) * M * N, MLA_E_ARG, "K-split workspace too small");
    MLA_REQUIRE(M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), MLA_E_SHAPE, "GEMM dimension overflow");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool r = relu != 0;
    // the double-buffered 128 x 128 kernel: its 64 KB of LDS let two workgroups share a CU, which hides the short ranges' prologue
    // (measured against the four-stage ring, same device: 36.9 vs 43.4 us at 10 240 rows, 15.6-19.7 vs 19.8 at 1 020)
    if (dtype == MLA_F32)
        return r ? launch<float, float, 4, 2, true>(a, lda, w, ldw, bias, out, ldo, M, N, K, s, splits, workspace)
                 : launch<float, float, 4, 2, false>(a, lda, w, ldw, bias, out, ldo, M, N, K, s, splits, workspace);
    if (out_dtype == MLA_F32)
        return r ? launch<mma::bf16_t, float, 4, 2, true>(a, lda, w, ldw, bias, out, ldo, M, N, K, s, splits, workspace)
                 : launch<mma::bf16_t, float, 4, 2, false>(a, lda, w, ldw, bias, out, ldo, M, N, K, s, splits, workspace);
    return r ? launch<mma::bf16_t, mma::bf16_t, 4, 2, true>(a, lda, w, ldw, bias, out, ldo, M, N, K, s, splits, workspace)
             : launch<mma::bf16_t, mma::bf16_t, 4, 2, false>(a, lda, w, ldw, bias, out, ldo, M, N, K, s, splits, workspace);
}

extern "C" int mla_linear(const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias, void* out,
                          int64_t ldo, int64_t M, int64_t N, int64_t K, int dtype, int out_dtype, int relu,
                          mla_stream_t stream) {
    MLA_REQUIRE(M >= 0 && N > 0 && K > 0, MLA_E_ARG, "bad GEMM shape %lld x %lld x %lld", (long long)M, (long long)N, (long long)K);
    if (M == 0) return MLA_OK;
    MLA_REQUIRE(a && w && out, MLA_E_ARG, "null GEMM operand");
    MLA_REQUIRE(dtype == MLA_F32 || dtype == MLA_BF16, MLA_E_DTYPE, "GEMM dtype %d", dtype);
    MLA_REQUIRE(out_dtype == MLA_F32 || (out_dtype == MLA_BF16 && dtype == MLA_BF16), MLA_E_DTYPE,
                "GEMM out dtype %d for compute dtype %d", out_dtype, dtype);
    const int per = dtype == MLA_F32 ? 4 : 8;
    MLA_REQUIRE(K % per == 0 && lda % per == 0 && ldw % per == 0 && lda >= K && ldw >= K && ldo >= N, MLA_E_SHAPE,
                "K / lda / ldw must be multiples of %d elements (16-byte rows): K %lld lda %lld ldw %lld ldo %lld", per,
                (long long)K, (long long)lda, (long long)ldw, (long long)ldo);
    MLA_REQUIRE(mla::aligned(a, 16) && mla::aligned(w, 16), MLA_E_ARG, "GEMM operands must be 16-byte aligned");
    MLA_REQUIRE(M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), MLA_E_SHAPE, "GEMM dimension overflow");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == MLA_F32) return dispatch<float, float>(a, lda, w, ldw, bias, out, ldo, M, N, K, relu != 0, s);
    if (out_dtype == MLA_F32) return dispatch<mma::bf16_t, float>(a, lda, w, ldw, bias, out, ldo, M, N, K, relu != 0, s);
    return dispatch<mma::bf16_t, mma::bf16_t>(a, lda, w, ldw, bias, out, ldo, M, N, K, relu != 0, s);
}

// "bf16x3" Linear (see MLA_BF16X3): a (M, 2K) = [hi | lo] per segment of `seg` columns, w (N, 3K) = [w_hi | w_lo | w_hi] per
// segment (mla_split_bf16x3), K = logical reduction length. out: float32 (M, N), or split bf16 (M, 2N) = [hi(N) | lo(N)].
extern "C" int mla_linear_bf16x3(const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias, void* out,
                                 int64_t ldo, int64_t M, int64_t N, int64_t K, int64_t seg, int out_dtype, int relu,
                                 mla_stream_t stream) {
    MLA_REQUIRE(M >= 0 && N > 0 && K > 0, MLA_E_ARG, "bad GEMM shape %lld x %lld x %lld", (long long)M, (long long)N, (long long)K);
    if (M == 0) return MLA_OK;
    MLA_REQUIRE(a && w && out, MLA_E_ARG, "null GEMM operand");
    MLA_REQUIRE(out_dtype == MLA_F32 || out_dtype == MLA_BF16X3, MLA_E_DTYPE, "split GEMM out dtype %d", out_dtype);
    MLA_REQUIRE(seg > 0 && seg % 64 == 0 && K % seg == 0 && lda >= 2 * K && ldw >= 3 * K && lda % 8 == 0 && ldw % 8 == 0 &&
                ldo >= (out_dtype == MLA_F32 ? N : 2 * N), MLA_E_SHAPE,
                "split GEMM: seg %lld must be a multiple of 64 dividing K %lld; lda %lld >= 2K, ldw %lld >= 3K", (long long)seg,
                (long long)K, (long long)lda, (long long)ldw);
    MLA_REQUIRE(mla::aligned(a, 16) && mla::aligned(w, 16), MLA_E_ARG, "GEMM operands must be 16-byte aligned");
    MLA_REQUIRE(M < (1ll << 31) && N < (1ll << 31) && 3 * K < (1ll << 31), MLA_E_SHAPE, "GEMM dimension overflow");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (out_dtype == MLA_F32)
        return dispatch<mma::bf16_t, float>(a, lda, w, ldw, bias, out, ldo, M, N, 3 * K, relu != 0, s, int(seg / 64), 0);
    return dispatch<mma::bf16_t, mma::bf16_t>(a, lda, w, ldw, bias, out, ldo, M, N, 3 * K, relu != 0, s, int(seg / 64), 1);
}
