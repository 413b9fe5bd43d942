// mg_common.h — shared host-side helpers for the C ABI (error string, launch checks).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/metagym_hip.h"

namespace mg {

// thread-local "last error" text behind mg_last_error()
char *error_buffer();
int set_error(int code, const char *fmt, ...);

inline int check_hip(hipError_t e, const char *what) {
    if (e == hipSuccess) return MG_OK;
    return set_error(-(int)e, "%s: %s", what, hipGetErrorString(e));
}

// peek (not sync): a launch error such as an invalid configuration surfaces here
inline int check_launch(const char *kernel) { return check_hip(hipGetLastError(), kernel); }

constexpr int WAVE = 64;  // gfx950 wavefront width

}  // namespace mg

#define MG_REQUIRE_PTR(p)                                                         \
    do {                                                                          \
        if ((p) == nullptr) return mg::set_error(MG_ERR_NULL_POINTER, "%s: %s is NULL", __func__, #p); \
    } while (0)
