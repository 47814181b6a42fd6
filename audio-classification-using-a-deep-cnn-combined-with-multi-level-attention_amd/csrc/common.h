// common.h -- shared host-side plumbing of libmla_hip.so (error reporting, launch checks).
#ifndef MLA_COMMON_H
#define MLA_COMMON_H

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/mla_hip.h"

namespace mla {

char* error_buffer();                       // thread-local, 512 bytes (defined in abi.hip)
int fail(int code, const char* fmt, ...);   // formats into error_buffer(), returns code

inline bool aligned(const void* p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

}  // namespace mla

#define MLA_REQUIRE(cond, code, ...) \
    do { if (!(cond)) return ::mla::fail((code), __VA_ARGS__); } while (0)

#define MLA_HIP_OK(expr)                                                                      \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return ::mla::fail(MLA_E_LAUNCH, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

#define MLA_LAUNCH_OK(what)                                                                   \
    do {                                                                                      \
        hipError_t e_ = hipGetLastError();                                                    \
        if (e_ != hipSuccess) return ::mla::fail(MLA_E_LAUNCH, "launch %s: %s", what, hipGetErrorString(e_)); \
    } while (0)

#endif
