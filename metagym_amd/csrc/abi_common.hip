// abi_common.hip — version / error-string entry points of libmetagym_hip.so.
#include <cstring>

#include "mg_common.h"
#include "mg_philox.h"

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

// Known-answer hook for the counter-based generator behind every fused auto-reset (tests/: Random123 vectors).
namespace {
__global__ void philox_selftest_kernel(const uint32_t *in, uint32_t *out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t *p = in + 6 * (size_t)i;
    uint32_t r[4];
    philox4x32_10(p[0], p[1], p[2], p[3], p[4], p[5], r);
    for (int k = 0; k < 4; ++k) out[4 * (size_t)i + k] = r[k];
}
}  // namespace

extern "C" int mg_selftest_philox(const uint32_t *ctr_key_d, uint32_t *out_d, int32_t n, void *stream) {
    MG_REQUIRE_PTR(ctr_key_d);
    MG_REQUIRE_PTR(out_d);
    if (n <= 0) return mg::set_error(MG_ERR_BAD_SIZE, "n=%d", n);
    mg::DeviceGuard guard(mg::device_of(out_d));
    hipLaunchKernelGGL(philox_selftest_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, ctr_key_d, out_d, n);
    return mg::check_launch("philox_selftest_kernel");
}
