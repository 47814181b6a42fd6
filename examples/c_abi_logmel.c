/* Plain-C host using libmla_hip.so through include/mla_hip.h only (no Python, no torch): 4 x 10 s of synthetic 16 kHz PCM ->
 * VGGish examples (vggish_input.waveform_to_examples, vggish_input.py:30-82) on the GPU, then a few numbers for eyeballing.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_abi_logmel.c -L<pkg> -lmla_hip \
 *       -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,<pkg> -Wl,-rpath,/opt/rocm/lib -o c_abi_logmel
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "mla_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_MLA(x) do { int rc_ = (x); if (rc_ != MLA_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, mla_last_error()); return 3; } } while (0)

int main(void) {
    const int64_t n_wave = 4, n_samples = 160000;
    int64_t frames = 0, examples = 0;
    CHECK_MLA(mla_logmel_counts(n_samples, &frames, &examples));
    printf("abi %d: %lld samples -> %lld STFT frames -> %lld examples per waveform\n", mla_abi_version(), (long long)n_samples,
           (long long)frames, (long long)examples);

    /* constant tables: built on the host once, uploaded by the caller */
    const int64_t tab_n = mla_logmel_table_floats();
    float* tab_h = (float*)malloc((size_t)tab_n * sizeof(float));
    CHECK_MLA(mla_logmel_build_tables(tab_h));

    float* pcm_h = (float*)malloc((size_t)(n_wave * n_samples) * sizeof(float));
    uint32_t s = 12345u;
    for (int64_t i = 0; i < n_wave * n_samples; ++i) { s = s * 1664525u + 1013904223u; pcm_h[i] = (float)((int32_t)s >> 8) * (1.0f / 8388608.0f) * 0.5f; }

    float *tab_d, *pcm_d, *out_d;
    const size_t out_n = (size_t)(n_wave * examples) * 96 * 64;
    CHECK_HIP(hipMalloc((void**)&tab_d, (size_t)tab_n * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&pcm_d, (size_t)(n_wave * n_samples) * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&out_d, out_n * sizeof(float)));
    CHECK_HIP(hipMemcpy(tab_d, tab_h, (size_t)tab_n * sizeof(float), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(pcm_d, pcm_h, (size_t)(n_wave * n_samples) * sizeof(float), hipMemcpyHostToDevice));

    /* asynchronous on the given stream (NULL = default stream); buffers are caller-owned */
    CHECK_MLA(mla_logmel_examples(pcm_d, MLA_F32, n_wave, n_samples, n_samples, tab_d, out_d, MLA_F32, NULL));
    CHECK_HIP(hipDeviceSynchronize());

    float* out_h = (float*)malloc(out_n * sizeof(float));
    CHECK_HIP(hipMemcpy(out_h, out_d, out_n * sizeof(float), hipMemcpyDeviceToHost));
    double sum = 0.0;
    for (size_t i = 0; i < out_n; ++i) sum += out_h[i];
    printf("examples (%lld, 96, 64): mean log-mel %.6f, first row %.4f %.4f %.4f ...\n", (long long)(n_wave * examples),
           sum / (double)out_n, out_h[0], out_h[1], out_h[2]);

    /* error behaviour: a waveform shorter than 240 samples is the reference's ValueError */
    const int rc = mla_logmel_examples(pcm_d, MLA_F32, 1, 200, 200, tab_d, out_d, MLA_F32, NULL);
    printf("n = 200 samples -> rc %d (%s)\n", rc, mla_last_error());
    return rc == MLA_E_SHORT ? 0 : 4;
}
