// logmel_core.h -- per-lane arithmetic of the fused log-mel kernel, written so that the
// SAME source runs (a) inside logmel.hip on gfx950 and (b) on the host, lane by lane,
// in csrc/logmel_hostsim.cpp (built with g++ by the CPU tests). The host simulation
// proves the index arithmetic (radix-16 x radix-16 FFT, real-FFT split, sparse mel) against
// the oracle without a GPU; the GPU parity tests then only have to catch execution-model
// bugs (barriers, LDS aliasing).
//
// Algorithm for ONE STFT frame, executed by a group of 16 lanes j = 0..15
// (reference: mel_features.py:71-92 stft_magnitude, :220-223 mel + log):
//
//   x[n], n < 400 windowed samples, zero-padded to 512 (mel_features.py:91-92)
//   z[m] = x[2m] + i x[2m+1], m < 256            (real FFT through a half-size complex FFT)
//   Z = FFT256(z) with 256 = 16 x 16:  m = 16 n1 + n2,  k = k1 + 16 k2
//     phase 1  lane n2 :  A[k1] = W256^(n2 k1) * sum_n1 z[16 n1 + n2] W16^(n1 k1)
//     exchange through LDS (16x16 transpose inside the group)
//     phase 2  lane k1 :  Z[k1 + 16 k2] = sum_n2 A_n2[k1] W16^(n2 k2)
//   2 X[k]     = (Z[k] + conj Z[256-k]) + w^k (Z[k] - conj Z[256-k]) / i,  w = e^(-2 pi i/512)
//   2 X[256-k] = conj( (Z[k] + conj Z[256-k]) - w^k (...) / i )
//     phase 3  stays in registers: lane j holds Z[j + 16 k2]; its mirror bins 256 - (j + 16 s) live in
//              lane (16 - j) & 15, registers 15 - s, and arrive through ONE 16-lane exchange per value;
//              lane j then produces |2X| for bins j + 16 s and 256 - j - 16 s, s = 0..7 (lane 0, which
//              pairs with itself one register off, is fixed up by a register rotation; its s = 0 slot
//              computes bin 128). Magnitudes go to LDS on top of the dead exchange buffer.
//   mel[b] = sum_k (M[k][b] / 2) |2 X[k]| over the band's bin range, widened to 16-byte aligned windows
//     phase 4  lane j owns bands {j, 31-j, 32+j, 63-j}; windows of 8 / 8 / 12 / 20 bins read as float4
//   out[b] = log(mel[b] + 0.01)
#ifndef MLA_LOGMEL_CORE_H
#define MLA_LOGMEL_CORE_H

#if defined(__HIPCC__)
#define MLA_HD __host__ __device__ __forceinline__
#else
#define MLA_HD inline
#endif

namespace logmel {

constexpr int kWin = 400, kHop = 160, kFft = 512, kBands = 64, kExFrames = 96;
constexpr int kN1 = 13;                    // non-zero first-stage inputs: 32*n1 + 2*n2 < 400
constexpr int kXchStride = 18;             // complex elements per exchange row (16 + 2 pad: rows stay 16-byte aligned and conflict-free for ds_read_b128)
constexpr int kSlot0 = 8, kSlot1 = 8, kSlot2 = 12, kSlot3 = 20;   // taps per band slot: 4-bin aligned windows
constexpr int kTaps = kSlot0 + kSlot1 + kSlot2 + kSlot3;          // 48

// table layout (float indices) ------------------------------------------------------
constexpr int kTabWindow = 0;              // 512 floats: Hann(400) then zeros
constexpr int kTabTw256 = 512;             // 256 x (cos, -sin)(2 pi m / 256)
constexpr int kTabTw512 = 1024;            // 129 x (cos, -sin)(2 pi k / 512), padded to 512
constexpr int kTabMelStart = 1536;         // 16 lanes x 4 slots, int32 first bin of the slot
constexpr int kMelRow = 52;                // per-lane weight row pitch: 52 j mod 64 banks distinct for ds_read_b128
constexpr int kPwRow = 16;                 // per-lane split twiddles: 8 x (re, im) of w^(j + 16 s) (lane 0, s = 0: w^128)
constexpr int kTabMelW = 1600;             // 16 lanes x kMelRow floats, weights / 2 (zero padded)
constexpr int kTabPw = kTabMelW + 16 * kMelRow;     // 16 lanes x kPwRow
constexpr int kTabFloats = kTabPw + 16 * kPwRow;    // 2688
constexpr int kLaneTabFloats = 16 * (kMelRow + kPwRow);   // contiguous [kTabMelW, kTabFloats): LDS-resident

MLA_HD int band_of(int lane, int slot) {
    return slot == 0 ? lane : slot == 1 ? 31 - lane : slot == 2 ? 32 + lane : 63 - lane;
}

// per-lane constants, loaded once per workgroup lifetime -------------------------------
struct LaneConsts {
    float tw_re[16], tw_im[16];            // W256^(j k1)
    int mel_start[4];
};

MLA_HD void load_consts(LaneConsts& c, const float* tab, int j) {
    _Pragma("unroll") for (int k1 = 0; k1 < 16; ++k1) {
        c.tw_re[k1] = tab[kTabTw256 + 2 * (j * k1)];
        c.tw_im[k1] = tab[kTabTw256 + 2 * (j * k1) + 1];
    }
    const int* starts = reinterpret_cast<const int*>(tab + kTabMelStart);
    _Pragma("unroll") for (int s = 0; s < 4; ++s) c.mel_start[s] = starts[4 * j + s];
}

// 4-point forward DFT (W4 = -i), in place on elements (0,1,2,3) of the argument list ------
MLA_HD void dft4(float& r0, float& i0, float& r1, float& i1, float& r2, float& i2, float& r3, float& i3) {
    const float t0r = r0 + r2, t0i = i0 + i2, t1r = r0 - r2, t1i = i0 - i2;
    const float t2r = r1 + r3, t2i = i1 + i3, t3r = r1 - r3, t3i = i1 - i3;
    r0 = t0r + t2r; i0 = t0i + t2i;
    r2 = t0r - t2r; i2 = t0i - t2i;
    r1 = t1r + t3i; i1 = t1i - t3r;       // t1 - i t3
    r3 = t1r - t3i; i3 = t1i + t3r;       // t1 + i t3
}
// same with the 4th input known to be zero (outputs written to all four)
MLA_HD void dft4_z3(float& r0, float& i0, float& r1, float& i1, float& r2, float& i2, float& r3, float& i3) {
    const float t0r = r0 + r2, t0i = i0 + i2, t1r = r0 - r2, t1i = i0 - i2;
    const float xr = r1, xi = i1;
    r0 = t0r + xr; i0 = t0i + xi;
    r2 = t0r - xr; i2 = t0i - xi;
    r1 = t1r + xi; i1 = t1i - xr;
    r3 = t1r - xi; i3 = t1i + xr;
}

// dft4 with input 2 given as u2 / R (R = sqrt(1/2)): the scaling rides in the first butterfly's FMAs
MLA_HD void dft4_s2(float& r0, float& i0, float& r1, float& i1, float& u2r, float& u2i, float& r3, float& i3) {
    constexpr float R = 0.70710678118654752440f;
    const float t0r = r0 + R * u2r, t0i = i0 + R * u2i, t1r = r0 - R * u2r, t1i = i0 - R * u2i;
    const float t2r = r1 + r3, t2i = i1 + i3, t3r = r1 - r3, t3i = i1 - i3;
    r0 = t0r + t2r; i0 = t0i + t2i;
    u2r = t0r - t2r; u2i = t0i - t2i;
    r1 = t1r + t3i; i1 = t1i - t3r;
    r3 = t1r - t3i; i3 = t1i + t3r;
}
// dft4 with inputs 1 and 3 given as u1 / R and u3 / R: t2 = R (u1 + u3), t3 = R (u1 - u3) are never formed, the scaling
// rides in the second butterfly's FMAs
MLA_HD void dft4_s13(float& r0, float& i0, float& u1r, float& u1i, float& r2, float& i2, float& u3r, float& u3i) {
    constexpr float R = 0.70710678118654752440f;
    const float t0r = r0 + r2, t0i = i0 + i2, t1r = r0 - r2, t1i = i0 - i2;
    const float sr = u1r + u3r, si = u1i + u3i, dr = u1r - u3r, di = u1i - u3i;
    r0 = t0r + R * sr; i0 = t0i + R * si;
    r2 = t0r - R * sr; i2 = t0i - R * si;
    u1r = t1r + R * di; u1i = t1i - R * dr;       // t1 - i t3
    u3r = t1r - R * di; u3i = t1i + R * dr;       // t1 + i t3
}

MLA_HD void cmul(float& r, float& i, float p, float q) {   // (r + i i) *= (p + i q)
    const float nr = r * p - i * q, ni = r * q + i * p;
    r = nr; i = ni;
}

// 16-point forward DFT of re/im[0..15] (index n = 4a + b). Output X[k], k = c + 4d, is left
// at element 4c + d (base-4 digit reversal); use dr16(k) to address it.
// LAST3_ZERO: inputs 13, 14, 15 are zero (first stage: only 13 non-zero n1).
MLA_HD constexpr int dr16(int k) { return 4 * (k & 3) + (k >> 2); }

template <bool LAST3_ZERO>
MLA_HD void dft16(float* re, float* im) {
    constexpr float C1 = 0.92387953251128673848f, S1 = 0.38268343236508978178f, R = 0.70710678118654752440f;
    // step 1: DFT4 over a for each b -> y_b[c] at element 4c + b
    dft4(re[0], im[0], re[4], im[4], re[8], im[8], re[12], im[12]);
    if (LAST3_ZERO) {
        dft4_z3(re[1], im[1], re[5], im[5], re[9], im[9], re[13], im[13]);
        dft4_z3(re[2], im[2], re[6], im[6], re[10], im[10], re[14], im[14]);
        dft4_z3(re[3], im[3], re[7], im[7], re[11], im[11], re[15], im[15]);
    } else {
        dft4(re[1], im[1], re[5], im[5], re[9], im[9], re[13], im[13]);
        dft4(re[2], im[2], re[6], im[6], re[10], im[10], re[14], im[14]);
        dft4(re[3], im[3], re[7], im[7], re[11], im[11], re[15], im[15]);
    }
    // step 2: y_b[c] *= W16^(b c), element 4c + b; W16^m = (cos, -sin)(2 pi m / 16)
    cmul(re[5], im[5], C1, -S1);                                   // b=1,c=1: m=1
    // the four twiddles of modulus-R components, (1 - i) R (m = 2) and -(1 + i) R (m = 6), are applied WITHOUT their factor
    // R = sqrt(1/2): it is folded into the FMAs of the following DFT4 (dft4_s2 / dft4_s13)
    { const float a = re[6], b = im[6]; re[6] = a + b; im[6] = b - a; }                  // m=2 (x R)
    cmul(re[7], im[7], S1, -C1);                                   // b=3,c=1: m=3
    { const float a = re[9], b = im[9]; re[9] = a + b; im[9] = b - a; }                  // b=1,c=2: m=2 (x R)
    { const float a = re[10], b = im[10]; re[10] = b; im[10] = -a; }                     // m=4: -i
    { const float a = re[11], b = im[11]; re[11] = b - a; im[11] = -(a + b); }           // m=6 (x R)
    cmul(re[13], im[13], S1, -C1);                                 // b=1,c=3: m=3
    { const float a = re[14], b = im[14]; re[14] = b - a; im[14] = -(a + b); }           // m=6 (x R)
    cmul(re[15], im[15], -C1, S1);                                 // b=3,c=3: m=9
    // step 3: DFT4 over b for each c: elements 4c .. 4c+3 -> X[c + 4d] at 4c + d
    dft4(re[0], im[0], re[1], im[1], re[2], im[2], re[3], im[3]);
    dft4_s2(re[4], im[4], re[5], im[5], re[6], im[6], re[7], im[7]);
    dft4_s13(re[8], im[8], re[9], im[9], re[10], im[10], re[11], im[11]);
    dft4_s2(re[12], im[12], re[13], im[13], re[14], im[14], re[15], im[15]);
}

// phase 1: lane j = n2. phase1_window: windowed samples of the lane, z[16 n1 + j] = (x w)[32 n1 + 2 j] +
// i (x w)[32 n1 + 2 j + 1]; `frame` points at the frame's first sample, `win` is the 400-point periodic Hann
// followed by zeros (n1 = 12 reaches index 415). The kernel does the same from registers (samples fetched
// straight from HBM/L2, one frame ahead).
MLA_HD void phase1_window(int j, const float* frame, const float* win, float* re, float* im) {
    _Pragma("unroll") for (int n1 = 0; n1 < 12; ++n1) {
        re[n1] = frame[32 * n1 + 2 * j] * win[32 * n1 + 2 * j];
        im[n1] = frame[32 * n1 + 2 * j + 1] * win[32 * n1 + 2 * j + 1];
    }
    if (j < 8) {      // samples 384 + 2j (+1) < 400; beyond: zero padding, never read
        re[12] = frame[384 + 2 * j] * win[384 + 2 * j];
        im[12] = frame[384 + 2 * j + 1] * win[384 + 2 * j + 1];
    } else {
        re[12] = 0.f; im[12] = 0.f;
    }
}
// phase1_fft: first radix-16 on re/im[0..12] (13..15 are zero padding), then writes A[k1] * W256^(j k1) to
// xch[k1 * 17 + j] (float2 = re, im).
MLA_HD void phase1_fft(const LaneConsts& c, int j, float* re, float* im, float* xch) {
    re[13] = re[14] = re[15] = 0.f;
    im[13] = im[14] = im[15] = 0.f;
    dft16<true>(re, im);
    _Pragma("unroll") for (int k1 = 0; k1 < 16; ++k1) {
        float r = re[dr16(k1)], i = im[dr16(k1)];
        if (k1) cmul(r, i, c.tw_re[k1], c.tw_im[k1]);
        xch[2 * (k1 * kXchStride + j)] = r;
        xch[2 * (k1 * kXchStride + j) + 1] = i;
    }
}
MLA_HD void phase1(const LaneConsts& c, int j, const float* frame, const float* win, float* xch) {
    float re[16], im[16];
    phase1_window(j, frame, win, re, im);
    phase1_fft(c, j, re, im, xch);
}

// phase 2a: lane j = k1 gathers its row of the exchange buffer (all lanes of the group must
// have finished phase 1). Split from 2b because the Z image written in 2b aliases xch.
MLA_HD void phase2_read(int j, const float* xch, float* re, float* im) {
    _Pragma("unroll") for (int n2 = 0; n2 < 16; ++n2) {
        re[n2] = xch[2 * (j * kXchStride + n2)];
        im[n2] = xch[2 * (j * kXchStride + n2) + 1];
    }
}
// phase 2b: second radix-16 in place; afterwards Z[j + 16 k2] sits in element dr16(k2) of lane j.
MLA_HD void phase2_fft(float* re, float* im) { dft16<false>(re, im); }

MLA_HD float fast_sqrt(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sqrtf(v);      // one v_sqrt_f32 (~1 ulp); sqrtf() expands to ~12 VALU of fix-ups
#else
    return __builtin_sqrtf(v);
#endif
}

// phase 3a: the 8 values lane (16 - j) & 15 needs from this lane: view[s] <-> register 15 - s, i.e.
// Z[j + 16 (15 - s)]. Lane 0 pairs with itself one register off (256 - 16 s = 16 (16 - s)) and uses its
// s = 0 slot for bin 128, so it rotates: view[s] = Z[16 (16 - s)] for s >= 1, view[0] = Z[128].
MLA_HD void phase3_view(int j, const float* re, const float* im, float* vr, float* vi) {
    _Pragma("unroll") for (int s = 0; s < 8; ++s) {
        const int gen = dr16(15 - s);                       // register 15 - s
        const int l0 = dr16(s == 0 ? 8 : 16 - s);           // lane 0: register 16 - s (s = 0: 8)
        vr[s] = j == 0 ? re[l0] : re[gen];
        vi[s] = j == 0 ? im[l0] : im[gen];
    }
}

// phase 3b: pr/pi = the partner lane's view (exchanged by the caller). |2 X| for bins j + 16 s (mag+)
// and 256 - j - 16 s (mag-), s = 0..7; lane 0, s = 0: bin 128 from Z[128] alone (written twice).
// pw: this lane's kPwRow split twiddles.
MLA_HD void phase3_pairs(int j, const float* re, const float* im, const float* pr, const float* pi, float* mag, const float* pw) {
    _Pragma("unroll") for (int s = 0; s < 8; ++s) {
        float a = re[dr16(s)], b = im[dr16(s)];
        if (s == 0) {                                       // lane 0: Z[128] (register 8) instead of Z[0]
            a = j == 0 ? re[dr16(8)] : a;
            b = j == 0 ? im[dr16(8)] : b;
        }
        const float p = pr[s], q = pi[s];
        const float er = a + p, ei = b - q;                 // 2 E[k] = Z[k] + conj Z[256-k]
        float orr = b + q, oi = p - a;                      // 2 O[k] = (Z[k] - conj Z[256-k]) / i
        cmul(orr, oi, pw[2 * s], pw[2 * s + 1]);            // w^k * 2 O[k]
        const float ur = er + orr, ui = ei + oi, vr = er - orr, vi = ei - oi;
        const int k = (s == 0 && j == 0) ? 128 : j + 16 * s;
        mag[k] = fast_sqrt(ur * ur + ui * ui);
        mag[256 - k] = fast_sqrt(vr * vr + vi * vi);
    }
}

// phase 4: four mel bands per lane over 16-byte aligned bin windows (weights zero outside the band),
// natural log with the reference's 0.01 offset. melw: this lane's kMelRow weights; mag 16-byte aligned.
MLA_HD void phase4(const LaneConsts& c, int j, const float* mag, const float* melw, float* out4) {
    constexpr int first[4] = {0, kSlot0, kSlot0 + kSlot1, kSlot0 + kSlot1 + kSlot2};
    constexpr int count[4] = {kSlot0, kSlot1, kSlot2, kSlot3};
    (void)j;
    _Pragma("unroll") for (int s = 0; s < 4; ++s) {
        // 16-byte aligned on both sides (window starts and row offsets are multiples of 4 floats):
        // explicit vector loads, hipcc cannot prove the alignment of mag + start and falls back to 4-byte reads
        typedef float v4f __attribute__((vector_size(16)));
        const v4f* m = reinterpret_cast<const v4f*>(mag + c.mel_start[s]);
        const v4f* w = reinterpret_cast<const v4f*>(melw + first[s]);
        float acc = 0.f;
        _Pragma("unroll") for (int t = 0; t < count[s] / 4; ++t) {
            const v4f mv = m[t], wv = w[t];
            acc += wv[0] * mv[0];
            acc += wv[1] * mv[1];
            acc += wv[2] * mv[2];
            acc += wv[3] * mv[3];
        }
#if defined(__HIP_DEVICE_COMPILE__)
        out4[s] = __builtin_amdgcn_logf(acc + 0.01f) * 0.69314718055994530942f;   // v_log_f32 (log2) * ln 2; argument >= 0.01
#else
        out4[s] = __builtin_logf(acc + 0.01f);
#endif
    }
}

}  // namespace logmel
#endif
