// Micro-benchmark (not product code): HBM floor of the quadrotor step's I/O pattern at N envs.
//   variant 0: the kernel's SoA layout, 23 dword/dwordx2 loads + stores per lane, obs via float4
//   variant 1: the same bytes packed as 16-byte vectors per lane (AoSoA)
//   variant 2: empty kernel (launch boundary only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void io_soa(const float *pos, const double *vel, const double *om, const float *pw, const float *R, const int *ct,
                       const float4 *act, float *opos, double *ovel, double *oom, float *opw, float *oR, int *oct,
                       float4 *obs, float *rew, unsigned char *done, int n) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float p[3], w[4], r[9]; double v[3], o[3];
    for (int c = 0; c < 3; ++c) { p[c] = pos[c * n + e]; v[c] = vel[c * n + e]; o[c] = om[c * n + e]; }
    for (int c = 0; c < 4; ++c) w[c] = pw[c * n + e];
    for (int c = 0; c < 9; ++c) r[c] = R[c * n + e];
    int k = ct[e]; float4 a = act[e];
    float s = a.x + a.y + a.z + a.w;
    for (int c = 0; c < 3; ++c) { opos[c * n + e] = p[c] + s; ovel[c * n + e] = v[c] + s; oom[c * n + e] = o[c] + s; }
    for (int c = 0; c < 4; ++c) opw[c * n + e] = w[c] + s;
    for (int c = 0; c < 9; ++c) oR[c * n + e] = r[c] + s;
    oct[e] = k + 1;
    for (int j = 0; j < 4; ++j) obs[e * 4 + j] = make_float4(p[0] + j, r[1], w[2], s);
    rew[e] = s; done[e] = (unsigned char)(k & 1);
}

__global__ void io_packed(const float4 *in, float4 *out, const float4 *act, float4 *obs, float *rew, unsigned char *done, int n) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
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
        for (int variant = 0; variant < 3; ++variant) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                for (int i = 0; i < 200; ++i) {
                    if (variant == 0)
                        io_soa<<<n / 256, 256>>>(f, d, d + 3 * n, f + 3 * n, f + 7 * n, (int *)(f + 16 * n), (float4 *)(f + 17 * n),
                                                 (float *)o, (double *)(o + 12 * n), (double *)(o + 36 * n), (float *)(o + 60 * n),
                                                 (float *)(o + 76 * n), (int *)(o + 112 * n), (float4 *)o2, (float *)(o2 + 64 * n),
                                                 (unsigned char *)(o2 + 68 * n), n);
                    else if (variant == 1)
                        io_packed<<<n / 256, 256>>>((float4 *)f, (float4 *)o, (float4 *)(f + 28 * n), (float4 *)o2, (float *)(o2 + 64 * n),
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
