// logmel.hip -- fused log-mel front-end for gfx950: PCM -> 0.96 s VGGish examples in one
// kernel (reference: vggish_input.py:30-82 -> mel_features.py:192-223).
//
// Roofline: HBM by bytes (algorithmic bytes per 0.96 s example: 15 360 new samples x 4 B for f32
// PCM, 2 B for int16, + 96 x 64 x 4 B out, 2 B for bf16 = 86 016 B), but what the kernel waits on
// is its own arithmetic and LDS transposes: ~600 f32 vector instructions per frame and lane
// (~1.5 MFLOP per example) and two LDS transposes per frame. Measured (rocprofv3 counters,
// profiles/): the vector pipe and the LDS are each ~50 % busy and the chip drops its clock to
// 1.9-2.1 GHz under this mix, so the design minimises instructions, LDS cycles and cache traffic
// per frame rather than chasing occupancy:
//
//   - persistent workgroups (256 threads = 16 groups of 16 lanes, 2 per CU) walk 32-frame
//     chunks grid-stride; after the table load there is NO workgroup barrier: a 16-lane group
//     never leaves its wave, so every hand-off is a wave-level fence (compiler ordering only --
//     a wave's LDS operations execute in issue order);
//   - each group owns TWO ADJACENT STFT frames per iteration. Its lanes fetch their sample pairs
//     straight from global memory (128-byte contiguous runs per group and instruction) one
//     iteration ahead, into registers. Frame B starts 160 = 5 x 32 samples after frame A, so B's
//     tap n1 is A's tap n1 + 5 under the same window value: 18 loads serve 26 taps. The remaining
//     frame overlap is served by L2; HBM sees each sample once (rocprofv3 FETCH_SIZE);
//   - radix-16 x radix-16 FFT with one LDS transpose per frame (rows of 18 complex: 16-byte
//     aligned and conflict-free for ds_read_b128), real-FFT split in registers via DPP,
//     magnitudes transposed through the dead exchange buffer, sparse triangular mel (<= 2 bands
//     per bin, 461 non-zeros, never the dense 257 x 64 product), log -- see logmel_core.h. The
//     two frames' phases are interleaved so one frame's LDS round trip hides behind the other's
//     arithmetic, and window / split twiddles / mel weights are read once per pair;
//   - finished rows are staged in the (dead) exchange buffers and leave as 16 B-per-lane stores,
//     2 KiB contiguous per wave, directly in (example, 96, 64) layout, so the 0.96 s windowing
//     (vggish_input.py:73-76) is free.
//
// Built with -fno-slp-vectorize (build.py): packed f32 has no throughput advantage on CDNA4 and
// hipcc pays ~100 v_mov shuffles per frame to form the register pairs.
#include <hip/hip_bf16.h>

#include "common.h"
#include "logmel_core.h"
#include "logmel_tables.h"

namespace {

using namespace logmel;

// Diagnostic build only (-DMLA_LOGMEL_STAMPS=1, scripts/build_variant.py): wall-clock (s_memrealtime, 100 MHz) start / end stamp of
// every workgroup, written to a buffer nothing else reads; read back through mla_debug_logmel_stamps(). The shipped kernel executes none.
#ifndef MLA_LOGMEL_STAMPS
#define MLA_LOGMEL_STAMPS 0
#endif
#if MLA_LOGMEL_STAMPS
__device__ unsigned long long g_logmel_stamps[256][10];     // per workgroup: kernel entry, end of each of its 8 waves, loop start
#endif

constexpr int kPair = 2;                                  // STFT frames per 16-lane group and iteration (adjacent frames)
constexpr int kChunk = 16 * kPair;                        // frames per workgroup iteration: a third of an example
constexpr int kThreads = 256;
constexpr int kWgPerCu = 2;                                // persistent workgroups per CU (LDS: 2 x 78 KB; 240 VGPRs)
constexpr int kXchFloats = 2 * 16 * kXchStride;           // 576 per frame in flight
constexpr int kWinFloats = 416;
constexpr int kPwPitch = 18;                              // split-twiddle row pitch: 16 rows on distinct bank pairs
constexpr int kU = 18;                                    // sample pairs per lane for two adjacent frames: 13 + 5
constexpr int kLdsFloats = kChunk * kXchFloats + 16 * kMelRow + 16 * kPwPitch + kWinFloats;
constexpr int kLdsBytes = kLdsFloats * 4;
static_assert(kWgPerCu * kLdsBytes <= 160 * 1024, "the persistent workgroups must fit one CU's LDS");
static_assert(kExFrames % kChunk == 0, "examples must split into whole chunks");
static_assert(kHop == 160 && kN1 == 13, "the frame-pair sample sharing assumes hop = 5 * 32 samples");

// group-local synchronisation. A 16-lane group never leaves its wave, and a wave's LDS
// instructions execute in issue order, so a compiler-level fence is sufficient (WAVE);
// BLOCK keeps a full workgroup barrier and exists to cross-check WAVE on hardware.
template <bool WAVE>
__device__ __forceinline__ void group_sync() {
    if (WAVE) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// value held by lane (16 - j) & 15 of the same 16-lane row: DPP row_mirror (j -> 15 - j) followed by
// row_ror:1 -- two VALU moves, no LDS round trip (verified on hardware: 0 15 14 ... 1)
__device__ __forceinline__ float from_partner(float v) {
    const int m = __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true);
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(m, 0x121, 0xf, 0xf, true));
}

// U[m] = (x[32 m + 2 j], x[32 m + 2 j + 1]), m < 18, relative to the first frame of the pair: frame A uses
// U[0..12], frame B (160 samples later) U[5..17]. U[17] exists for j < 8 only (B's samples 384 .. 399).
template <typename InT> struct Samples;

template <> struct Samples<float> {
    f32x2 u[kU];
    template <bool VEC>
    __device__ __forceinline__ void fetch(const float* frame, int j) {
        const float* p = frame + 2 * j;
        _Pragma("unroll") for (int m = 0; m < kU - 1; ++m) {
            u[m] = VEC ? *reinterpret_cast<const f32x2*>(p + 32 * m) : f32x2{p[32 * m], p[32 * m + 1]};
        }
        u[kU - 1] = f32x2{0.f, 0.f};
        if (j < 8) u[kU - 1] = VEC ? *reinterpret_cast<const f32x2*>(p + 32 * (kU - 1)) : f32x2{p[32 * (kU - 1)], p[32 * (kU - 1) + 1]};
    }
    __device__ __forceinline__ f32x2 pair(int m) const { return u[m]; }
};

template <> struct Samples<int16_t> {
    uint32_t u[kU];
    template <bool VEC>
    __device__ __forceinline__ void fetch(const int16_t* frame, int j) {
        const uint16_t* p = reinterpret_cast<const uint16_t*>(frame) + 2 * j;
        _Pragma("unroll") for (int m = 0; m < kU - 1; ++m) {
            u[m] = VEC ? *reinterpret_cast<const uint32_t*>(p + 32 * m) : (p[32 * m] | (uint32_t(p[32 * m + 1]) << 16));
        }
        u[kU - 1] = 0u;
        if (j < 8) u[kU - 1] = VEC ? *reinterpret_cast<const uint32_t*>(p + 32 * (kU - 1)) : (p[32 * (kU - 1)] | (uint32_t(p[32 * (kU - 1) + 1]) << 16));
    }
    // int16 -> float in [-1, 1): x / 32768 exactly (vggish_input.py:98)
    __device__ __forceinline__ f32x2 pair(int m) const {
        constexpr float k = 1.0f / 32768.0f;
        return f32x2{k * float(int16_t(u[m] & 0xFFFFu)), k * float(int16_t(u[m] >> 16))};
    }
};

struct ChunkMap {
    int64_t wave_stride;
    int chunks_per_wave;           // examples_per_wave * 3
    __device__ __forceinline__ void locate(int64_t c, int64_t& sample0, int64_t& row0) const {
        const int64_t w = c / chunks_per_wave, r = c - w * chunks_per_wave;
        sample0 = w * wave_stride + r * (kChunk * kHop);
        row0 = c * kChunk;
    }
};

// the lane's row of the exchange buffer: 16 complex values, 16-byte aligned
__device__ __forceinline__ void read_row(const float* p, float* re, float* im) {
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(p + 4 * i);
        re[2 * i] = q.x; im[2 * i] = q.y; re[2 * i + 1] = q.z; im[2 * i + 1] = q.w;
    }
}

// phase 3 of one frame: real-FFT split, magnitudes into the (dead) exchange buffer
template <bool WAVE>
__device__ __forceinline__ void split_frame(int j, const float* re, const float* im, float* xg, const float* pw) {
    float vr[8], vi[8], pr[8], pi[8];
    phase3_view(j, re, im, vr, vi);
    _Pragma("unroll") for (int s = 0; s < 8; ++s) {
        pr[s] = from_partner(vr[s]);
        pi[s] = from_partner(vi[s]);
    }
    phase3_pairs(j, re, im, pr, pi, xg, pw);
}

// phase 4 for both frames of the pair: logmel_core.h's phase4 arithmetic (same summation order per band), each
// weight vector read once. All 36 LDS reads are issued before the first FMA (the FFT registers are dead here), so
// the LDS latency is paid once, and the eight accumulators advance in turn (no dependent FMA chains).
__device__ __forceinline__ void mel_pair(const LaneConsts& c, const float* mag_a, const float* mag_b, const float* melw,
                                         float* oa, float* ob) {
    constexpr int first[4] = {0, kSlot0, kSlot0 + kSlot1, kSlot0 + kSlot1 + kSlot2};
    constexpr int count[4] = {kSlot0 / 4, kSlot1 / 4, kSlot2 / 4, kSlot3 / 4};
    f32x4 w[kTaps / 4], a[kTaps / 4], b[kTaps / 4];
    _Pragma("unroll") for (int s = 0; s < 4; ++s) {
        _Pragma("unroll") for (int t = 0; t < count[s]; ++t) {
            w[first[s] / 4 + t] = *reinterpret_cast<const f32x4*>(melw + first[s] + 4 * t);
            a[first[s] / 4 + t] = *reinterpret_cast<const f32x4*>(mag_a + c.mel_start[s] + 4 * t);
            b[first[s] / 4 + t] = *reinterpret_cast<const f32x4*>(mag_b + c.mel_start[s] + 4 * t);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    float acc_a[4] = {0.f, 0.f, 0.f, 0.f}, acc_b[4] = {0.f, 0.f, 0.f, 0.f};
    _Pragma("unroll") for (int t = 0; t < kSlot3 / 4; ++t) {
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {
            _Pragma("unroll") for (int s = 0; s < 4; ++s) {
                if (t < count[s]) {
                    acc_a[s] += w[first[s] / 4 + t][e] * a[first[s] / 4 + t][e];
                    acc_b[s] += w[first[s] / 4 + t][e] * b[first[s] / 4 + t][e];
                }
            }
        }
    }
    _Pragma("unroll") for (int s = 0; s < 4; ++s) {
        oa[s] = __builtin_amdgcn_logf(acc_a[s] + 0.01f) * 0.69314718055994530942f;
        ob[s] = __builtin_amdgcn_logf(acc_b[s] + 0.01f) * 0.69314718055994530942f;
    }
}

template <typename OutT>
__device__ __forceinline__ void store_piece(OutT* dst, f32x4 v) {
    if constexpr (sizeof(OutT) == 4) {
        *reinterpret_cast<f32x4*>(dst) = v;
    } else {
        const __hip_bfloat16 b0 = __float2bfloat16(v.x), b1 = __float2bfloat16(v.y), b2 = __float2bfloat16(v.z), b3 = __float2bfloat16(v.w);
        auto bits = [](const __hip_bfloat16& h) { return uint32_t(*reinterpret_cast<const uint16_t*>(&h)); };
        *reinterpret_cast<u32x2*>(dst) = u32x2{bits(b0) | (bits(b1) << 16), bits(b2) | (bits(b3) << 16)};
    }
}

// One frame pair of a 16-lane group: window (samples already in `smp`), prefetch of the group's next pair, both FFTs, split,
// mel, log, and the two finished rows to dst / dst + 64. Shared by the dynamic (shipped) and the static (cross-check) kernel.
template <typename InT, typename OutT, bool VEC, bool WAVE>
__device__ __forceinline__ void pair_step(const LaneConsts& c, int j, Samples<InT>& smp, const InT* next_frame, float* xa, float* xb,
                                          const float* s_win, const float* melw, const float* pw, OutT* dst) {
    float ra[16], ia[16], rb[16], ib[16];
    {   // window: frame B's tap n1 is sample pair U[n1 + 5] under the SAME window value as A's tap n1
        _Pragma("unroll") for (int n1 = 0; n1 < kN1; ++n1) {
            const f32x2 w = *reinterpret_cast<const f32x2*>(s_win + 32 * n1 + 2 * j);   // zeros beyond 399
            const f32x2 a = smp.pair(n1), b = smp.pair(n1 + 5);
            ra[n1] = a.x * w.x; ia[n1] = a.y * w.y;
            rb[n1] = b.x * w.x; ib[n1] = b.y * w.y;
        }
    }
    if (next_frame) smp.template fetch<VEC>(next_frame, j);       // the next pair's samples fly while this one computes
    phase1_fft(c, j, ra, ia, xa);
    group_sync<WAVE>();
    phase1_fft(c, j, rb, ib, xb);
    group_sync<WAVE>();
    // both rows and the split twiddles are requested together; A's second FFT starts when ITS eight reads are back
    read_row(xa + 2 * (j * kXchStride), ra, ia);
    read_row(xb + 2 * (j * kXchStride), rb, ib);
    float pwl[16];
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {
        const f32x2 q = *reinterpret_cast<const f32x2*>(pw + 2 * i);
        pwl[2 * i] = q.x; pwl[2 * i + 1] = q.y;
    }
    __builtin_amdgcn_sched_barrier(0);
    group_sync<WAVE>();                              // every lane has requested its rows: both buffers are dead
    phase2_fft(ra, ia);
    split_frame<WAVE>(j, ra, ia, xa, pwl);           // magnitudes of A overwrite its buffer
    phase2_fft(rb, ib);
    split_frame<WAVE>(j, rb, ib, xb, pwl);
    group_sync<WAVE>();
    float oa[4], ob[4];
    mel_pair(c, xa, xb, melw, oa, ob);
    // finished rows -> floats 256..319 of the (dead) buffers, then one 16-byte piece per lane and row:
    // a wave stores its eight rows as 2 KiB contiguous
    _Pragma("unroll") for (int s = 0; s < 4; ++s) { xa[256 + band_of(j, s)] = oa[s]; xb[256 + band_of(j, s)] = ob[s]; }
    group_sync<WAVE>();
    {
        const f32x4 va = *reinterpret_cast<const f32x4*>(xa + 256 + 4 * j);
        const f32x4 vb = *reinterpret_cast<const f32x4*>(xb + 256 + 4 * j);
        store_piece<OutT>(dst + 4 * j, va);
        store_piece<OutT>(dst + kBands + 4 * j, vb);
    }
    group_sync<WAVE>();                 // the buffers are rewritten by the next pair
}

// Static kernel (cross-check build, MLA_LOGMEL_SYNC=block, and the round-1 schedule): 256 threads, two workgroups per CU, chunks of
// 32 frames dealt grid-stride; WAVE = false keeps a full workgroup barrier at every hand-off.
template <typename InT, typename OutT, bool VEC, bool WAVE>
__global__ __launch_bounds__(kThreads, kWgPerCu) void logmel_kernel(const InT* __restrict__ pcm, ChunkMap map,
                                                              int64_t n_chunks, const float* __restrict__ tab,
                                                              OutT* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_xch = smem;
    float* s_mel = s_xch + kChunk * kXchFloats;   // per-lane mel weights
    float* s_pw = s_mel + 16 * kMelRow;           // per-lane split twiddles
    float* s_win = s_pw + 16 * kPwPitch;          // Hann window + zero tail

    const int t = threadIdx.x, g = t >> 4, j = t & 15;
    float* xa = s_xch + (kPair * g) * kXchFloats;
    float* xb = xa + kXchFloats;

    LaneConsts c;
    load_consts(c, tab, j);
    for (int i = t; i < 16 * kMelRow; i += kThreads) s_mel[i] = tab[kTabMelW + i];
    for (int i = t; i < 16 * kPwRow; i += kThreads) s_pw[(i >> 4) * kPwPitch + (i & 15)] = tab[kTabPw + i];
    for (int i = t; i < kWinFloats; i += kThreads) s_win[i] = tab[kTabWindow + i];
    const float* melw = s_mel + kMelRow * j;
    const float* pw = s_pw + kPwPitch * j;
    __syncthreads();                        // tables visible

    Samples<InT> smp;
    int64_t chunk = blockIdx.x, sample0, row0;
    if (chunk < n_chunks) {
        map.locate(chunk, sample0, row0);
        smp.template fetch<VEC>(pcm + sample0 + (kPair * g) * kHop, j);
    }
    for (; chunk < n_chunks; chunk += gridDim.x) {
        map.locate(chunk, sample0, row0);
        const int64_t next = chunk + gridDim.x;
        const InT* next_frame = nullptr;
        if (next < n_chunks) {
            int64_t ns, nr;
            map.locate(next, ns, nr);
            next_frame = pcm + ns + (kPair * g) * kHop;
        }
        pair_step<InT, OutT, VEC, WAVE>(c, j, smp, next_frame, xa, xb, s_win, melw, pw, out + (row0 + kPair * g) * kBands);
    }
}

// Dynamic kernel (shipped): ONE 512-thread workgroup per CU; work item = 8 frames = one wave-iteration (4 groups x 2 frames). A
// workgroup owns a contiguous range of items and its 8 waves PULL them from an LDS counter. Why: with equal static shares the two
// waves of a SIMD do not advance equally -- vector issue is arbitrated by age, the older wave runs ~1.5x faster, finishes its share
// early and leaves its partner alone on the SIMD for the last third of the kernel (a single wave issues at half the pair's rate):
// wall-clock stamps of the static kernel showed the first-dispatched workgroup of every CU done at 225-233 us and the second at
// 335-355 us of a 358 us launch (profiles/r02_frontend_counters.txt). No workgroup barrier after the table load, no global state.
constexpr int kThreadsDyn = 512;
constexpr int kItemFrames = 4 * kPair;                     // frames per wave-iteration
constexpr int kLdsFloatsDyn = (kThreadsDyn / 16) * kPair * kXchFloats + 16 * kMelRow + 16 * kPwPitch + kWinFloats + 4;
constexpr int kLdsBytesDyn = kLdsFloatsDyn * 4;
static_assert(kLdsBytesDyn <= 160 * 1024 && kChunk % kItemFrames == 0, "dynamic kernel: LDS / item size");

template <typename InT, typename OutT, bool VEC>
__global__ __launch_bounds__(kThreadsDyn, 1) void logmel_dyn_kernel(const InT* __restrict__ pcm, ChunkMap map, int64_t n_items,
                                                                    const float* __restrict__ tab, OutT* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_xch = smem;
    float* s_mel = s_xch + (kThreadsDyn / 16) * kPair * kXchFloats;
    float* s_pw = s_mel + 16 * kMelRow;
    float* s_win = s_pw + 16 * kPwPitch;
    int* s_next = reinterpret_cast<int*>(s_win + kWinFloats);

    const int t = threadIdx.x, g = t >> 4, gl = g & 3, j = t & 15, lane = t & 63;
#if MLA_LOGMEL_STAMPS
    const unsigned long long stamp_entry = __builtin_amdgcn_s_memrealtime();
#endif
    float* xa = s_xch + (kPair * g) * kXchFloats;
    float* xb = xa + kXchFloats;

    LaneConsts c;
    load_consts(c, tab, j);
    for (int i = t; i < 16 * kMelRow; i += kThreadsDyn) s_mel[i] = tab[kTabMelW + i];
    for (int i = t; i < 16 * kPwRow; i += kThreadsDyn) s_pw[(i >> 4) * kPwPitch + (i & 15)] = tab[kTabPw + i];
    for (int i = t; i < kWinFloats; i += kThreadsDyn) s_win[i] = tab[kTabWindow + i];
    // this workgroup's contiguous share of the items (consecutive frames: the 2.5x frame overlap stays in this XCD's L2)
    const int lo = int(n_items * int64_t(blockIdx.x) / int64_t(gridDim.x)), hi = int(n_items * (int64_t(blockIdx.x) + 1) / int64_t(gridDim.x));
    if (t == 0) *s_next = lo;
    const float* melw = s_mel + kMelRow * j;
    const float* pw = s_pw + kPwPitch * j;
    __syncthreads();                        // the only workgroup barrier: tables and the counter visible

    auto pull = [&]() {                      // wave-uniform: one lane takes the next item of the workgroup's share
        int v = 0;
        if (lane == 0) v = atomicAdd(s_next, 1);
        return __builtin_amdgcn_readfirstlane(v);
    };
    constexpr int kQ = kChunk / kItemFrames;                  // items per 32-frame chunk of the chunk map
    auto locate_item = [&](int item, int64_t& sample, int64_t& row) {     // wave-uniform: first sample / output row of the item
        int64_t sample0, row0;
        map.locate(item / kQ, sample0, row0);
        const int f = (item % kQ) * kItemFrames;
        sample = sample0 + int64_t(f) * kHop;
        row = row0 + f;
    };
    const int lane_frame = kPair * gl;                        // this group's frame pair inside the item
    Samples<InT> smp;
    int cur = pull();
    int64_t sample = 0, row = 0;
    if (cur < hi) {
        locate_item(cur, sample, row);
        smp.template fetch<VEC>(pcm + sample + lane_frame * kHop, j);
    }
#if MLA_LOGMEL_STAMPS
    const unsigned long long stamp0 = __builtin_amdgcn_s_memrealtime();
#endif
    while (cur < hi) {
        const int nxt = pull();
        int64_t nsample = 0, nrow = 0;
        const InT* next_frame = nullptr;
        if (nxt < hi) {
            locate_item(nxt, nsample, nrow);
            next_frame = pcm + nsample + lane_frame * kHop;
        }
        pair_step<InT, OutT, VEC, true>(c, j, smp, next_frame, xa, xb, s_win, melw, pw, out + (row + lane_frame) * kBands);
        cur = nxt;
        row = nrow;
    }
#if MLA_LOGMEL_STAMPS
    if (lane == 0 && blockIdx.x < 256) {
        if (t == 0) { g_logmel_stamps[blockIdx.x][0] = stamp_entry; g_logmel_stamps[blockIdx.x][9] = stamp0; }
        g_logmel_stamps[blockIdx.x][1 + (t >> 6)] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

template <typename InT, typename OutT>
int launch(const void* pcm, int64_t n_wave, int64_t wave_stride, int64_t examples, const float* tables,
           void* out, bool wave_sync, bool static_wave, hipStream_t stream) {
    const int64_t n_chunks = n_wave * examples * (kExFrames / kChunk);
    if (n_chunks == 0) return MLA_OK;
    const bool vec = mla::aligned(pcm, 2 * sizeof(InT)) && (wave_stride % 2 == 0);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int64_t grid = n_chunks < kWgPerCu * int64_t(cus) ? n_chunks : kWgPerCu * int64_t(cus);
    ChunkMap map{wave_stride, int(examples * (kExFrames / kChunk))};
    auto go = [&](auto kern) -> int {
        MLA_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
        hipLaunchKernelGGL(kern, dim3(unsigned(grid)), dim3(kThreads), kLdsBytes, stream,
                           static_cast<const InT*>(pcm), map, n_chunks, tables, static_cast<OutT*>(out));
        MLA_LAUNCH_OK("logmel_kernel");
        return MLA_OK;
    };
    if (wave_sync && !static_wave) {            // shipped: one 512-thread workgroup per CU, waves pull 8-frame items from an LDS counter
        const int64_t n_items = n_chunks * (kChunk / kItemFrames);
        MLA_REQUIRE(n_items <= 0x7fffffff, MLA_E_SHAPE, "too many frames for one launch (%lld items)", (long long)n_items);
        const int64_t wgs = (n_items + 7) / 8 < cus ? (n_items + 7) / 8 : cus;
        auto god = [&](auto kern) -> int {
            MLA_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytesDyn));
            hipLaunchKernelGGL(kern, dim3(unsigned(wgs)), dim3(kThreadsDyn), kLdsBytesDyn, stream,
                               static_cast<const InT*>(pcm), map, n_items, tables, static_cast<OutT*>(out));
            MLA_LAUNCH_OK("logmel_dyn_kernel");
            return MLA_OK;
        };
        return vec ? god(logmel_dyn_kernel<InT, OutT, true>) : god(logmel_dyn_kernel<InT, OutT, false>);
    }
    if (static_wave) return vec ? go(logmel_kernel<InT, OutT, true, true>) : go(logmel_kernel<InT, OutT, false, true>);
    return vec ? go(logmel_kernel<InT, OutT, true, false>) : go(logmel_kernel<InT, OutT, false, false>);
}

}  // namespace

#if MLA_LOGMEL_STAMPS
extern "C" int mla_debug_logmel_stamps(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_logmel_stamps), sizeof(g_logmel_stamps)) == hipSuccess ? 0 : -4;
}
#endif

extern "C" int mla_logmel_counts(int64_t n_samples, int64_t* stft_frames, int64_t* examples) {
    // mel_features.py:42: 1 + int(floor((n - 400) / 160)); negative counts raise in as_strided.
    MLA_REQUIRE(n_samples >= 0, MLA_E_ARG, "n_samples %lld < 0", (long long)n_samples);
    const int64_t d = n_samples - kWin;
    const int64_t fl = d >= 0 ? d / kHop : -((-d + kHop - 1) / kHop);      // floor division
    const int64_t frames = 1 + fl;
    MLA_REQUIRE(frames >= 0, MLA_E_SHORT, "waveform of %lld samples gives a negative frame count (reference raises ValueError)",
                (long long)n_samples);
    const int64_t d2 = frames - kExFrames;
    const int64_t fl2 = d2 >= 0 ? d2 / kExFrames : -((-d2 + kExFrames - 1) / kExFrames);
    const int64_t ex = 1 + fl2;
    if (stft_frames) *stft_frames = frames;
    if (examples) *examples = ex > 0 ? ex : 0;
    return MLA_OK;
}

extern "C" int64_t mla_logmel_table_floats(void) { return kTabFloats; }

extern "C" int mla_logmel_build_tables(float* host_out) {
    MLA_REQUIRE(host_out != nullptr, MLA_E_ARG, "host_out is null");
    const int rc = build_tables(host_out);
    MLA_REQUIRE(rc == 0, MLA_E_SHAPE, "mel band structure does not fit the kernel's padded slots (%d)", rc);
    return MLA_OK;
}

extern "C" int mla_logmel_reference_tables(double* host_window400, double* host_mel_257x64) {
    MLA_REQUIRE(host_window400 && host_mel_257x64, MLA_E_ARG, "null output");
    hann400(host_window400);
    mel_dense(host_mel_257x64);
    return MLA_OK;
}

extern "C" int mla_logmel_examples(const void* pcm, int pcm_dtype, int64_t n_wave, int64_t n_samples,
                                   int64_t wave_stride, const float* tables, void* out, int out_dtype,
                                   mla_stream_t stream) {
    MLA_REQUIRE(n_wave >= 0 && wave_stride >= n_samples, MLA_E_ARG, "bad n_wave %lld / stride %lld < n_samples %lld",
                (long long)n_wave, (long long)wave_stride, (long long)n_samples);
    int64_t frames = 0, examples = 0;
    const int rc = mla_logmel_counts(n_samples, &frames, &examples);
    if (rc != MLA_OK) return rc;
    if (n_wave == 0 || examples == 0) return MLA_OK;
    MLA_REQUIRE(pcm && tables && out, MLA_E_ARG, "null pcm/tables/out");
    MLA_REQUIRE(mla::aligned(out, 16) && mla::aligned(tables, 4), MLA_E_ARG, "out must be 16-byte aligned");
    MLA_REQUIRE(pcm_dtype == MLA_F32 || pcm_dtype == MLA_I16, MLA_E_DTYPE, "pcm_dtype %d", pcm_dtype);
    MLA_REQUIRE(out_dtype == MLA_F32 || out_dtype == MLA_BF16, MLA_E_DTYPE, "out_dtype %d", out_dtype);
    MLA_REQUIRE(mla::aligned(pcm, pcm_dtype == MLA_F32 ? 4 : 2), MLA_E_ARG, "pcm misaligned for its dtype");
    // MLA_LOGMEL_SYNC: "block" = cross-check build of the group sync (static kernel, full barriers); "static" = the round-1
    // schedule (static shares, wave-level fences) for A/B against the shipped dynamic kernel
    const char* env = getenv("MLA_LOGMEL_SYNC");
    const bool wave_sync = !(env && env[0] == 'b'), static_wave = env && env[0] == 's';
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (pcm_dtype == MLA_F32) {
        return out_dtype == MLA_F32 ? launch<float, float>(pcm, n_wave, wave_stride, examples, tables, out, wave_sync, static_wave, s)
                                    : launch<float, __hip_bfloat16>(pcm, n_wave, wave_stride, examples, tables, out, wave_sync, static_wave, s);
    }
    return out_dtype == MLA_F32 ? launch<int16_t, float>(pcm, n_wave, wave_stride, examples, tables, out, wave_sync, static_wave, s)
                                : launch<int16_t, __hip_bfloat16>(pcm, n_wave, wave_stride, examples, tables, out, wave_sync, static_wave, s);
}
