// logmel_hostsim.cpp -- TEST HARNESS (never part of libmla_hip.so): runs logmel_core.h's
// per-lane phases on the host, one 16-lane group at a time, with plain arrays standing in
// for LDS. Built with g++ by tests/test_logmel_hostsim.py to check the FFT / real-split /
// sparse-mel index arithmetic and the constant tables against the oracle without a GPU.
#include <cstdint>
#include <vector>

#include "logmel_core.h"
#include "logmel_tables.h"

using namespace logmel;

extern "C" int hostsim_table_floats() { return kTabFloats; }
extern "C" int hostsim_build_tables(float* tab) { return build_tables(tab); }
extern "C" void hostsim_reference_tables(double* win, double* mel) { hann400(win); mel_dense(mel); }

// pcm: n_samples floats; out: examples * 96 * 64 floats. Returns the number of examples.
extern "C" int64_t hostsim_examples(const float* pcm, int64_t n_samples, float* out) {
    if (n_samples < kWin) return 0;
    const int64_t frames = 1 + (n_samples - kWin) / kHop;
    const int64_t examples = frames / kExFrames;
    std::vector<float> tab(kTabFloats);
    if (build_tables(tab.data()) != 0) return -1;
    LaneConsts c[16];
    for (int j = 0; j < 16; ++j) load_consts(c[j], tab.data(), j);
    std::vector<float> xch(2 * 16 * kXchStride);
    float re[16][16], im[16][16], vr[16][8], vi[16][8];
    for (int64_t f = 0; f < examples * kExFrames; ++f) {
        const float* frame = pcm + f * kHop;
        for (int j = 0; j < 16; ++j) phase1(c[j], j, frame, tab.data() + kTabWindow, xch.data());
        for (int j = 0; j < 16; ++j) phase2_read(j, xch.data(), re[j], im[j]);
        for (int j = 0; j < 16; ++j) phase2_fft(re[j], im[j]);
        for (int j = 0; j < 16; ++j) phase3_view(j, re[j], im[j], vr[j], vi[j]);
        float* mag = xch.data();                                   // magnitudes alias the dead exchange buffer
        for (int j = 0; j < 16; ++j) {                             // the 16-lane exchange: partner = (16 - j) & 15
            const int partner = (16 - j) & 15;
            phase3_pairs(j, re[j], im[j], vr[partner], vi[partner], mag, tab.data() + kTabPw + kPwRow * j);
        }
        for (int j = 0; j < 16; ++j) {
            float o[4];
            phase4(c[j], j, mag, tab.data() + kTabMelW + kMelRow * j, o);
            for (int s = 0; s < 4; ++s) out[f * kBands + band_of(j, s)] = o[s];
        }
    }
    return examples;
}
