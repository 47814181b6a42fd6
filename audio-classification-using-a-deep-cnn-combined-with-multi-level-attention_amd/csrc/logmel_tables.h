// logmel_tables.h -- host-side construction of the fused kernel's constant tables in double
// precision, following the reference formulas:
//   periodic Hann            mel_features.py:48-68
//   HTK mel filterbank       mel_features.py:100-111, :114-189 (64 bands, 125..7500 Hz,
//                            257 bins over linspace(0, 8000, 257), DC row zeroed)
// plus the FFT twiddles. Shared by logmel.hip (mla_logmel_build_tables) and the host
// simulation used by the CPU tests.
#ifndef MLA_LOGMEL_TABLES_H
#define MLA_LOGMEL_TABLES_H

#include <cmath>
#include <cstring>
#include <vector>

#include "logmel_core.h"

namespace logmel {

inline double hz_to_mel(double hz) { return 1127.0 * std::log(1.0 + hz / 700.0); }

inline void hann400(double* w) {
    const double pi = 3.14159265358979323846;
    for (int n = 0; n < kWin; ++n) w[n] = 0.5 - 0.5 * std::cos(2.0 * pi / kWin * n);
}

// dense (257 x 64) row-major mel matrix
inline void mel_dense(double* m) {
    const int bins = kFft / 2 + 1;
    const double nyq = 8000.0;
    std::vector<double> bin_mel(bins);
    for (int k = 0; k < bins; ++k) {
        // numpy.linspace(0, nyq, bins): start + k * step with step = nyq / (bins - 1)
        const double hz = (k == bins - 1) ? nyq : k * (nyq / (bins - 1));
        bin_mel[k] = hz_to_mel(hz);
    }
    const double lo = hz_to_mel(125.0), hi = hz_to_mel(7500.0);
    std::vector<double> edges(kBands + 2);
    for (int i = 0; i < kBands + 2; ++i)
        edges[i] = (i == kBands + 1) ? hi : lo + i * ((hi - lo) / (kBands + 1));
    for (int k = 0; k < bins; ++k)
        for (int b = 0; b < kBands; ++b) {
            const double l = edges[b], c = edges[b + 1], u = edges[b + 2];
            const double rise = (bin_mel[k] - l) / (c - l), fall = (u - bin_mel[k]) / (u - c);
            double w = rise < fall ? rise : fall;
            if (w < 0.0) w = 0.0;
            m[k * kBands + b] = (k == 0) ? 0.0 : w;
        }
}

// returns 0 on success, negative if the band structure does not fit the padded slots
inline int build_tables(float* tab) {
    const double pi = 3.14159265358979323846;
    std::memset(tab, 0, sizeof(float) * kTabFloats);
    double w[kWin];
    hann400(w);
    for (int n = 0; n < kWin; ++n) tab[kTabWindow + n] = (float)w[n];
    for (int m = 0; m < 256; ++m) {
        tab[kTabTw256 + 2 * m] = (float)std::cos(2.0 * pi * m / 256.0);
        tab[kTabTw256 + 2 * m + 1] = (float)(-std::sin(2.0 * pi * m / 256.0));
    }
    for (int k = 0; k <= 128; ++k) {
        tab[kTabTw512 + 2 * k] = (float)std::cos(2.0 * pi * k / 512.0);
        tab[kTabTw512 + 2 * k + 1] = (float)(-std::sin(2.0 * pi * k / 512.0));
    }
    std::vector<double> mel(257 * kBands);
    mel_dense(mel.data());
    const int count[4] = {kSlot0, kSlot1, kSlot2, kSlot3};
    int* starts = reinterpret_cast<int*>(tab + kTabMelStart);
    for (int lane = 0; lane < 16; ++lane) {
        int first = 0;
        for (int s = 0; s < 4; ++s) {
            const int b = band_of(lane, s);
            int k0 = -1, k1 = -1;
            for (int k = 0; k < 257; ++k)
                if (mel[k * kBands + b] != 0.0) { if (k0 < 0) k0 = k; k1 = k; }
            if (k0 < 1 || k1 > 255) return -1;
            for (int k = k0; k <= k1; ++k)
                if (mel[k * kBands + b] == 0.0) return -2;          // support must be contiguous
            const int start = k0 / 4 * 4;                           // 16-byte aligned window (float4 reads)
            if (k1 - start + 1 > count[s] || start + count[s] > 256 || start < 4) return -1;
            starts[4 * lane + s] = start;
            for (int t = 0; t < count[s]; ++t) {
                const int k = start + t;
                const double v = (k >= k0 && k <= k1) ? mel[k * kBands + b] : 0.0;
                tab[kTabMelW + kMelRow * lane + first + t] = (float)(0.5 * v);   // kernel holds |2X|
            }
            first += count[s];
        }
        for (int i = 0; i < 8; ++i) {
            const int k = (lane == 0 && i == 0) ? 128 : lane + 16 * i;     // lane 0 uses its s = 0 slot for bin 128
            tab[kTabPw + kPwRow * lane + 2 * i] = tab[kTabTw512 + 2 * k];
            tab[kTabPw + kPwRow * lane + 2 * i + 1] = tab[kTabTw512 + 2 * k + 1];
        }
    }
    return 0;
}

}  // namespace logmel
#endif
