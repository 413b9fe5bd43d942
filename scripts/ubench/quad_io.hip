// Micro-benchmark (not product code): HBM floor of the quadrotor step's I/O pattern at N envs.
//   variant 0: the kernel's SoA layout, 23 dword/dwordx2 loads + stores per lane, obs via float4
//   variant 1: the same bytes packed as 16-byte vectors per lane (AoSoA)
//   variant 2: empty kernel (launch boundary only)
//   REMAP=1: blockIdx -> env block so that each XCD (blockIdx % 8) owns one contiguous eighth of the envs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int PAD = 1088;   // elements
template <int REMAP> __device__ __forceinline__ int env_block() {
    if (REMAP && (gridDim.x & 7) == 0) return (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    return blockIdx.x;
}

// REMAP doubles as MODE here: 0 loads + stores, 1 loads only (one dependent store), 2 stores only, 3 loads + nontemporal stores
template <int REMAP> __global__ void io_soa(const float *pos, const double *vel, const double *om, const float *pw, const float *R, const int *ct,
                       const float4 *act, float *opos, double *ovel, double *oom, float *opw, float *oR, int *oct,
                       float4 *obs, float *rew, unsigned char *done, int n) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float p[3] = {1, 2, 3}, w[4] = {1, 2, 3, 4}, r[9] = {1, 2, 3, 4, 5, 6, 7, 8, 9}; double v[3] = {1, 2, 3}, o[3] = {1, 2, 3};
    int k = e; float s = (float)e;
    if (REMAP != 2) {
        for (int c = 0; c < 3; ++c) { p[c] = pos[c * n + e]; v[c] = vel[c * n + e]; o[c] = om[c * n + e]; }
        for (int c = 0; c < 4; ++c) w[c] = pw[c * n + e];
        for (int c = 0; c < 9; ++c) r[c] = R[c * n + e];
        k = ct[e]; float4 a = act[e];
        s = a.x + a.y + a.z + a.w;
    }
    if (REMAP == 1) {
        float acc = s + k;
        for (int c = 0; c < 3; ++c) acc += p[c] + (float)v[c] + (float)o[c];
        for (int c = 0; c < 4; ++c) acc += w[c];
        for (int c = 0; c < 9; ++c) acc += r[c];
        rew[e] = acc;
        return;
    }
#define ST(ptr, val) do { if (REMAP == 3) __builtin_nontemporal_store((val), (ptr)); else *(ptr) = (val); } while (0)
    for (int c = 0; c < 3; ++c) { ST(&opos[c * n + e], p[c] + s); ST(&ovel[c * n + e], v[c] + s); ST(&oom[c * n + e], o[c] + s); }
    for (int c = 0; c < 4; ++c) ST(&opw[c * n + e], w[c] + s);
    for (int c = 0; c < 9; ++c) ST(&oR[c * n + e], r[c] + s);
    ST(&oct[e], k + 1);
    for (int j = 0; j < 4; ++j) { float *q = (float *)&obs[e * 4 + j]; ST(q, p[0] + j); ST(q + 1, r[1]); ST(q + 2, w[2]); ST(q + 3, s); }
    ST(&rew[e], s); ST(&done[e], (unsigned char)(k & 1));
}

template <int REMAP> __global__ void io_packed(const float4 *in, float4 *out, const float4 *act, float4 *obs, float *rew, unsigned char *done, int n) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    if (REMAP) n += PAD / 4;
    float4 a = act[e];
    float s = a.x + a.y + a.z + a.w;
    float4 x[7];
    for (int c = 0; c < 7; ++c) x[c] = in[c * n + e];       // 112 B ~ the 116 B of state
    for (int c = 0; c < 7; ++c) { x[c].x += s; out[c * n + e] = x[c]; }
    for (int j = 0; j < 4; ++j) obs[e * 4 + j] = make_float4(x[0].x + j, x[1].y, x[2].z, s);
    rew[e] = s; done[e] = (unsigned char)1;
}

__global__ void empty_k(int n) {}

int main() {
    for (int n : {65536, 1048576}) {
        size_t B = (size_t)n * 256;
        char *buf; hipMalloc(&buf, B * 4);
        hipMemset(buf, 0, B * 4);
        float *f = (float *)buf; double *d = (double *)(buf + B); char *o = buf + 2 * B; char *o2 = buf + 3 * B;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int variant = 0; variant < 7; ++variant) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                for (int i = 0; i < 200; ++i) {
                    if (variant == 5 || variant == 6) {
                        auto kern = variant == 5 ? io_soa<2> : io_soa<3>;
                        kern<<<n / 256, 256>>>(f, d, d + 3 * n, f + 3 * n, f + 7 * n, (int *)(f + 16 * n), (float4 *)(f + 17 * n),
                                                 (float *)o, (double *)(o + 12 * n), (double *)(o + 36 * n), (float *)(o + 60 * n),
                                                 (float *)(o + 76 * n), (int *)(o + 112 * n), (float4 *)o2, (float *)(o2 + 64 * n),
                                                 (unsigned char *)(o2 + 68 * n), n);
                    } else if (variant == 3)
                        io_soa<1><<<n / 256, 256>>>(f, d, d + 3 * n, f + 3 * n, f + 7 * n, (int *)(f + 16 * n), (float4 *)(f + 17 * n),
                                                 (float *)o, (double *)(o + 12 * n), (double *)(o + 36 * n), (float *)(o + 60 * n),
                                                 (float *)(o + 76 * n), (int *)(o + 112 * n), (float4 *)o2, (float *)(o2 + 64 * n),
                                                 (unsigned char *)(o2 + 68 * n), n);
                    else if (variant == 4)
                        io_packed<1><<<n / 256, 256>>>((float4 *)f, (float4 *)o, (float4 *)(f + 28 * n), (float4 *)o2, (float *)(o2 + 64 * n),
                                                    (unsigned char *)(o2 + 68 * n), n);
                    else if (variant == 0)
                        io_soa<0><<<n / 256, 256>>>(f, d, d + 3 * n, f + 3 * n, f + 7 * n, (int *)(f + 16 * n), (float4 *)(f + 17 * n),
                                                 (float *)o, (double *)(o + 12 * n), (double *)(o + 36 * n), (float *)(o + 60 * n),
                                                 (float *)(o + 76 * n), (int *)(o + 112 * n), (float4 *)o2, (float *)(o2 + 64 * n),
                                                 (unsigned char *)(o2 + 68 * n), n);
                    else if (variant == 1)
                        io_packed<0><<<n / 256, 256>>>((float4 *)f, (float4 *)o, (float4 *)(f + 28 * n), (float4 *)o2, (float *)(o2 + 64 * n),
                                                    (unsigned char *)(o2 + 68 * n), n);
                    else
                        empty_k<<<n / 256, 256>>>(n);
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            printf("n=%d variant=%d  %.2f us/launch\n", n, variant, ms / 200 * 1000);
        }
        hipFree(buf);
    }
    return 0;
}
