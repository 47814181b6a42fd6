// logmel.hip -- fused log-mel front-end for gfx950: PCM -> 0.96 s VGGish examples in one
// kernel (reference: vggish_input.py:30-82 -> mel_features.py:192-223).
//
// Roofline: HBM. Algorithmic bytes per 0.96 s example: 15 360 new samples x 4 B (f32 PCM;
// 2 B for int16) + 96 x 64 x 4 B out (2 B for bf16) = 86 016 B; the arithmetic (~1.5 MFLOP
// per example, f32 VALU) sits at ~17 FLOP/B, just under the f32 ridge, so the kernel is
// written to keep both the VALU and the memory pipe busy:
//
//   - persistent workgroups (256 threads, 2 per CU by LDS) walk 32-frame chunks grid-stride;
//     per-lane constants (window, twiddles, sparse mel weights: ~110 VGPRs) load once;
//   - a chunk's PCM span (5 360 samples, frames overlap 2.5x) is read from HBM exactly once,
//     16 B per lane, coalesced, into registers while the previous chunk computes
//     (issue-early / write-late staging), then parked in LDS where the 32 frames are cut;
//   - each 16-lane group owns one STFT frame at a time: radix-16 x radix-16 FFT with one LDS
//     transpose, real-FFT split, sparse triangular mel (<= 2 bands per bin, 461 non-zeros,
//     never the dense 257 x 64 product), log -- see logmel_core.h;
//   - finished rows are staged in LDS and leave as 16 B-per-lane contiguous stores, directly
//     in (example, 96, 64) layout, so the 0.96 s windowing (vggish_input.py:73-76) is free.
#include <hip/hip_bf16.h>

#include "common.h"
#include "logmel_core.h"
#include "logmel_tables.h"

namespace {

using namespace logmel;

constexpr int kChunk = 16;                                // STFT frames per chunk (6 per example): one frame per 16-lane group
constexpr int kThreads = 256;
constexpr int kSpan = (kChunk - 1) * kHop + kWin;         // 5360 samples per chunk
constexpr int kXchFloats = 2 * 16 * kXchStride;           // 544 per 16-lane group
constexpr int kWinFloats = 416;                           // Hann(400) + zeros up to the last index phase 1 touches
constexpr int kLdsFloats = kSpan + 16 * kXchFloats + kLaneTabFloats + kWinFloats;   // magnitudes and output rows alias the exchange buffer
constexpr int kLdsBytes = kLdsFloats * 4;                 // 52 032 B -> 3 workgroups per CU
static_assert(3 * kLdsBytes <= 160 * 1024, "three workgroups must fit one CU's LDS");
constexpr int kStageVec = (kSpan / 4 + kThreads - 1) / kThreads;   // 3 float4 per thread

static_assert(kSpan % 8 == 0, "chunk span must be vector-loadable");
static_assert(kExFrames % kChunk == 0, "examples must split into whole chunks");

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

// clang ext-vectors (not HIP's float4/uint4 structs): those keep staging arrays in scratch
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// value held by lane (16 - j) & 15 of the same 16-lane row: DPP row_mirror (j -> 15 - j) followed by
// row_ror:1 -- two VALU moves, no LDS round trip (verified on hardware: 0 15 14 ... 1)
__device__ __forceinline__ float from_partner(float v) {
    const int m = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, m, 0x121, 0xf, 0xf, true));
}

template <typename InT> struct Stage;

template <> struct Stage<float> {
    f32x4 v[kStageVec];
    template <bool VEC>
    __device__ __forceinline__ void load(const float* src, int t) {
        _Pragma("unroll") for (int r = 0; r < kStageVec; ++r) {
            // clamped, unconditional loads keep the staging registers out of scratch
            const int q = min(t + kThreads * r, kSpan / 4 - 1);
            if (VEC) {
                v[r] = reinterpret_cast<const f32x4*>(src)[q];
            } else {
                v[r] = f32x4{src[4 * q], src[4 * q + 1], src[4 * q + 2], src[4 * q + 3]};
            }
        }
    }
    __device__ __forceinline__ void store(float* s_pcm, int t) const {
        _Pragma("unroll") for (int r = 0; r < kStageVec; ++r) {
            const int q = t + kThreads * r;
            if (q < kSpan / 4) reinterpret_cast<f32x4*>(s_pcm)[q] = v[r];
        }
    }
};

constexpr int kStageVecI16 = (kSpan / 8 + kThreads - 1) / kThreads;   // 2 x (8 int16) per thread

template <> struct Stage<int16_t> {
    u32x4 v[kStageVecI16];
    template <bool VEC>
    __device__ __forceinline__ void load(const int16_t* src, int t) {
        _Pragma("unroll") for (int r = 0; r < kStageVecI16; ++r) {
            const int q = min(t + kThreads * r, kSpan / 8 - 1);
            if (VEC) {
                v[r] = reinterpret_cast<const u32x4*>(src)[q];
            } else {
                const uint16_t* s = reinterpret_cast<const uint16_t*>(src) + 8 * q;
                v[r] = u32x4{s[0] | (uint32_t(s[1]) << 16), s[2] | (uint32_t(s[3]) << 16),
                             s[4] | (uint32_t(s[5]) << 16), s[6] | (uint32_t(s[7]) << 16)};
            }
        }
    }
    // int16 -> float in [-1, 1): x / 32768 exactly (vggish_input.py:98)
    __device__ __forceinline__ void store(float* s_pcm, int t) const {
        constexpr float k = 1.0f / 32768.0f;
        _Pragma("unroll") for (int r = 0; r < kStageVecI16; ++r) {
            const int q = t + kThreads * r;
            if (q < kSpan / 8) {
                f32x4 lo, hi;
                lo.x = k * float(int16_t(v[r].x & 0xFFFFu)); lo.y = k * float(int16_t(v[r].x >> 16));
                lo.z = k * float(int16_t(v[r].y & 0xFFFFu)); lo.w = k * float(int16_t(v[r].y >> 16));
                hi.x = k * float(int16_t(v[r].z & 0xFFFFu)); hi.y = k * float(int16_t(v[r].z >> 16));
                hi.z = k * float(int16_t(v[r].w & 0xFFFFu)); hi.w = k * float(int16_t(v[r].w >> 16));
                reinterpret_cast<f32x4*>(s_pcm)[2 * q] = lo;
                reinterpret_cast<f32x4*>(s_pcm)[2 * q + 1] = hi;
            }
        }
    }
};

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const __hip_bfloat16 a = __float2bfloat16(lo), b = __float2bfloat16(hi);
    return uint32_t(*reinterpret_cast<const uint16_t*>(&a)) | (uint32_t(*reinterpret_cast<const uint16_t*>(&b)) << 16);
}

// chunk c of the job -> (first PCM sample, first output row)
struct ChunkMap {
    int64_t wave_stride;
    int chunks_per_wave;           // examples_per_wave * 3
    __device__ __forceinline__ void locate(int64_t c, int64_t& sample0, int64_t& row0) const {
        const int64_t w = c / chunks_per_wave, r = c - w * chunks_per_wave;
        sample0 = w * wave_stride + r * (kChunk * kHop);
        row0 = c * kChunk;          // rows are (wave, example, frame)-major == chunk-major
    }
};

template <typename InT, typename OutT, bool VEC, bool WAVE>
__global__ __launch_bounds__(kThreads, 3) void logmel_kernel(const InT* __restrict__ pcm, ChunkMap map,
                                                              int64_t n_chunks, const float* __restrict__ tab,
                                                              OutT* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_pcm = smem;
    float* s_xch = s_pcm + kSpan;
    float* s_tab = s_xch + 16 * kXchFloats;  // per-lane mel weights + split twiddles
    float* s_win = s_tab + kLaneTabFloats;   // Hann window, read by every group (broadcast)

    const int t = threadIdx.x, g = t >> 4, j = t & 15;
    float* xg = s_xch + g * kXchFloats;

    LaneConsts c;
    load_consts(c, tab, j);
    for (int i = t; i < kLaneTabFloats; i += kThreads) s_tab[i] = tab[kTabMelW + i];
    for (int i = t; i < kWinFloats; i += kThreads) s_win[i] = tab[kTabWindow + i];
    const float* melw = s_tab + kMelRow * j;
    const float* pw = s_tab + 16 * kMelRow + kPwRow * j;
    // (the first __syncthreads() of the chunk loop orders these writes before any read)

    Stage<InT> stage;
    int64_t chunk = blockIdx.x, sample0, row0;
    if (chunk < n_chunks) {
        map.locate(chunk, sample0, row0);
        stage.template load<VEC>(pcm + sample0, t);
    }
    for (; chunk < n_chunks; chunk += gridDim.x) {
        map.locate(chunk, sample0, row0);
        stage.store(s_pcm, t);
        __syncthreads();
        const int64_t next = chunk + gridDim.x;
        if (next < n_chunks) {              // in flight while this chunk computes
            int64_t ns, nr;
            map.locate(next, ns, nr);
            stage.template load<VEC>(pcm + ns, t);
        }
        {
            phase1(c, j, s_pcm + g * kHop, s_win, xg);
            group_sync<WAVE>();
            float re[16], im[16];
            phase2_read(j, xg, re, im);
            phase2_fft(re, im);
            // real-FFT split in registers: the mirror bins live in lane (16 - j) & 15 of this group
            float vr[8], vi[8], pr[8], pi[8];
            phase3_view(j, re, im, vr, vi);
            _Pragma("unroll") for (int s = 0; s < 8; ++s) {
                pr[s] = from_partner(vr[s]);
                pi[s] = from_partner(vi[s]);
            }
            group_sync<WAVE>();                 // every lane has read its exchange row: the buffer is dead
            phase3_pairs(j, re, im, pr, pi, xg, pw);        // magnitudes overwrite it (floats 0..255)
            group_sync<WAVE>();
            float o[4];
            phase4(c, j, xg, melw, o);
            // finished row -> floats 256..319 of the group's (dead) exchange buffer
            _Pragma("unroll") for (int s = 0; s < 4; ++s) xg[256 + band_of(j, s)] = o[s];
            group_sync<WAVE>();
            // wave w holds rows 4w..4w+3 of this 16-row slab: 256 floats = one float4 per lane,
            // so the store needs no cross-wave barrier and is 1 KiB contiguous per wave.
            const int64_t o0 = row0 * kBands;
            const f32x4 v = *reinterpret_cast<const f32x4*>(s_xch + (t >> 4) * kXchFloats + 256 + 4 * (t & 15));
            if constexpr (sizeof(OutT) == 4) {
                reinterpret_cast<f32x4*>(out + o0)[t] = v;
            } else {
                reinterpret_cast<u32x2*>(out + o0)[t] = u32x2{pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
            }
            group_sync<WAVE>();                 // the buffer is rewritten by the next frame
        }
        __syncthreads();                    // every frame cut from s_pcm before it is overwritten
    }
}

template <typename InT, typename OutT>
int launch(const void* pcm, int64_t n_wave, int64_t wave_stride, int64_t examples, const float* tables,
           void* out, bool wave_sync, hipStream_t stream) {
    const int64_t n_chunks = n_wave * examples * (kExFrames / kChunk);
    if (n_chunks == 0) return MLA_OK;
    constexpr int64_t vec_elems = 16 / sizeof(InT);
    const bool vec = mla::aligned(pcm, 16) && (wave_stride % vec_elems == 0);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int64_t grid = n_chunks < 3 * int64_t(cus) ? n_chunks : 3 * int64_t(cus);   // 3 persistent workgroups per CU
    ChunkMap map{wave_stride, int(examples * (kExFrames / kChunk))};
    auto go = [&](auto kern) -> int {
        MLA_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
        hipLaunchKernelGGL(kern, dim3(unsigned(grid)), dim3(kThreads), kLdsBytes, stream,
                           static_cast<const InT*>(pcm), map, n_chunks, tables, static_cast<OutT*>(out));
        MLA_LAUNCH_OK("logmel_kernel");
        return MLA_OK;
    };
    if (vec) return wave_sync ? go(logmel_kernel<InT, OutT, true, true>) : go(logmel_kernel<InT, OutT, true, false>);
    return wave_sync ? go(logmel_kernel<InT, OutT, false, true>) : go(logmel_kernel<InT, OutT, false, false>);
}

}  // namespace

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
    const char* env = getenv("MLA_LOGMEL_SYNC");            // "block" = cross-check build of the group sync
    const bool wave_sync = !(env && env[0] == 'b');
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (pcm_dtype == MLA_F32) {
        return out_dtype == MLA_F32 ? launch<float, float>(pcm, n_wave, wave_stride, examples, tables, out, wave_sync, s)
                                    : launch<float, __hip_bfloat16>(pcm, n_wave, wave_stride, examples, tables, out, wave_sync, s);
    }
    return out_dtype == MLA_F32 ? launch<int16_t, float>(pcm, n_wave, wave_stride, examples, tables, out, wave_sync, s)
                                : launch<int16_t, __hip_bfloat16>(pcm, n_wave, wave_stride, examples, tables, out, wave_sync, s);
}
