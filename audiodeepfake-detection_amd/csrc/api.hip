// Error reporting and version of libafd_hip.
#include "afd_common.h"
#include "../../include/afd_hip.h"

namespace afd {

char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace afd

extern "C" const char* afd_last_error(void) { return afd::error_buffer(); }
extern "C" int afd_version(void) { return 1; }
