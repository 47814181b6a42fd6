// mma_core.h -- shared device pieces of the MFMA kernels (conv.hip, gemm.hip).
//
// Geometry shared by the bf16 and the f32 ("parity") modes: every LDS tile row is 128 bytes
// = 8 chunks of 16 B (64 bf16 or 32 f32 channels of one pixel / one weight row). One "k-step"
// consumes 4 chunks (64 B): one v_mfma_f32_16x16x32_bf16, or four v_mfma_f32_16x16x4_f32.
//
// Fragment maps (cdna_hip_programming.md section 3):
//   16x16x32 bf16 : lane l holds A[row l&15][k = 8(l>>4) + j], B[k = 8(l>>4) + j][col l&15]
//   16x16x4  f32  : lane l holds A[row l&15][k = l>>4],        B[k = l>>4][col l&15]
//   C/D (both)    : col = l&15, row = 4(l>>4) + reg
// so in both modes lane l reads chunk q = l>>4 of the k-step from tile row l&15 with ONE
// ds_read_b128. For f32 the four MFMAs of a k-step take elements 0..3 of that chunk, i.e. the
// GEMM K order inside 16 channels is permuted (k = 4 s + q <-> channel 4 q + s) identically
// for A and B, which leaves the sum unchanged.
//
// Bank conflicts: rows are 128 B, two per 256-B LDS bank row (16 slots of 16 B). A ds_read_b128
// lane group is {rows 0-3, 12-15 at chunk q} + {rows 4-11 at chunk q^1}: the two classes differ in
// chunk bit 0, so the swizzle leaves bit 0 alone and XORs bits 1-2 with (row >> 1) & 3. Inside a
// class the 8 rows are two runs of 4 that are 12 apart = consecutive mod 8, so (row & 1, (row>>1)&3)
// are 8 distinct slot pairs for ANY starting row -- i.e. also for the kx-shifted taps of the
// convolution (a swizzle on all three bits is conflict-free only for unshifted reads; measured
// 22 % SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE before this fix).
#ifndef MLA_MMA_CORE_H
#define MLA_MMA_CORE_H

#include <hip/hip_runtime.h>

#include <cstdint>

namespace mma {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct bf16_t { uint16_t bits; };          // storage type of bf16 tensors (2 bytes)

constexpr int kRowBytes = 128;
template <typename T> struct Elem;
template <> struct Elem<bf16_t> { static constexpr int kPerChunk = 8, kPerRow = 64; };
template <> struct Elem<float> { static constexpr int kPerChunk = 4, kPerRow = 32; };

__device__ __forceinline__ uint16_t f2bf(float f) {     // round-to-nearest-even, NaN preserved
    __bf16 h = static_cast<__bf16>(f);
    return __builtin_bit_cast(uint16_t, h);
}
__device__ __forceinline__ float bf2f(uint16_t b) { return __builtin_bit_cast(float, uint32_t(b) << 16); }
// two floats -> one dword of bf16 (lo | hi << 16): ONE v_cvt_pk_bf16_f32 (converting singly costs a cvt each plus a shift/or pair)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}

template <typename T> __device__ __forceinline__ void store_elem(T* p, float v);
template <> __device__ __forceinline__ void store_elem<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void store_elem<bf16_t>(bf16_t* p, float v) { p->bits = f2bf(v); }
template <typename T> __device__ __forceinline__ float load_elem(const T* p);
template <> __device__ __forceinline__ float load_elem<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float load_elem<bf16_t>(const bf16_t* p) { return bf2f(p->bits); }

// one k-step of a 16x16 tile: acc += A(16 x 4 chunks) * B(4 chunks x 16)
template <typename T> __device__ __forceinline__ void mma_step(const u32x4& a, const u32x4& b, f32x4& acc);
template <> __device__ __forceinline__ void mma_step<bf16_t>(const u32x4& a, const u32x4& b, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma_step<float>(const u32x4& a, const u32x4& b, f32x4& acc) {
    const f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.x, fb.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.y, fb.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.z, fb.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.w, fb.w, acc, 0, 0, 0);
}

__device__ __forceinline__ u32x4 lds_read16(const char* base, int byte_off) {
    return *reinterpret_cast<const u32x4*>(base + byte_off);
}
__device__ __forceinline__ void lds_write16(char* base, int byte_off, u32x4 v) {
    *reinterpret_cast<u32x4*>(base + byte_off) = v;
}
__device__ __forceinline__ u32x4 zero16() { return u32x4{0u, 0u, 0u, 0u}; }

// byte offset of (row, chunk) in a plain [rows][128 B] tile with the XOR swizzle
__device__ __forceinline__ int tile_off(int row, int chunk) {
    return row * kRowBytes + 16 * (chunk ^ (((row >> 1) & 3) << 1));
}

}  // namespace mma
#endif
