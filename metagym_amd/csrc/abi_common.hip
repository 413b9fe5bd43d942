// abi_common.hip — version / error-string entry points of libmetagym_hip.so.
#include <cstring>

#include "mg_common.h"

namespace mg {

char *error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}

int set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace mg

extern "C" int mg_abi_version(void) { return MG_ABI_VERSION; }
extern "C" const char *mg_last_error(void) { return mg::error_buffer(); }
extern "C" const char *mg_target_arch(void) { return "gfx950"; }
