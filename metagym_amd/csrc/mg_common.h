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
#define MG_MAX_DEVICES 64
namespace mg {

// HIP device that owns a device allocation (-1: not a device pointer / lookup failed; the launch then runs on
// the current device as before and faults loudly if that is wrong).
inline int device_of(const void *p) {
    hipPointerAttribute_t a;
    if (p == nullptr || hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    return a.device;
}

// Kernels launch on the calling thread's CURRENT device; a caller holding tensors of cuda:1 while cuda:0 is
// current would otherwise launch cuda:0 kernels on cuda:1 pointers. Switch for the duration of one ABI call.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int dev) {
        if (dev < 0) return;
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

// Workgroup b runs on XCD b % 8 (the hardware deals workgroups round-robin, whatever their cost). A frame's
// cost follows its view, views follow the task and the pose, and callers lay tasks out periodically (env e
// -> task e % T): with the identity mapping each XCD would render the same few tasks for the whole
// launch and the launch would last as long as the unluckiest XCD (maze3d: +35 % measured with agents near
// their start cells). Inside every aligned group of 8 envs the env -> XCD assignment is therefore rotated by a
// different amount per group, so any period in e spreads over all XCDs. Same 8 frames per group of 8
// workgroups as before, so locality is unchanged; a ragged last group keeps the identity.
__device__ __forceinline__ int env_of_block(int b, int n) {
    const int q = b >> 3, x = b & 7;
    if (((q + 1) << 3) > n) return b;
    return (q << 3) | ((x + q + (q >> 3) + (q >> 6) + (q >> 9)) & 7);
}

}  // namespace mg

#define MG_REQUIRE_PTR(p)                                                         \
    do {                                                                          \
        if ((p) == nullptr) return mg::set_error(MG_ERR_NULL_POINTER, "%s: %s is NULL", __func__, #p); \
    } while (0)
