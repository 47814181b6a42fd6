// abi.hip -- version / error accessors of the C ABI (include/mla_hip.h).
#include <cstring>

#include "common.h"

namespace mla {

char* error_buffer() {
    static thread_local char buf[512] = "";
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace mla

extern "C" int mla_abi_version(void) { return 1; }
extern "C" const char* mla_last_error(void) { return mla::error_buffer(); }
