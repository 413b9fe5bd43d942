// Micro-benchmark (not product code): issue cost of the f64 VALU instructions the maze renderer's pixel loop is made of, on gfx950.
// Each kernel runs one instruction 8 x 64 x ITER times per wave on 8 independent register sets (no dependent chain), with enough
// waves (8 per SIMD) to keep every SIMD's VALU port busy; reported: shader cycles per wave-instruction per SIMD (4 = full rate for a
// 64-wide wave on a 16-lane SIMD).   hipcc --offload-arch=gfx950 -O3 f64_rates.hip -o f64_rates && ./f64_rates
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define DEF_KERNEL_D2D(NAME, INSTR)                                                                     \
    __global__ void NAME(double *out, int iters) {                                                      \
        double a[8], b = 1.0000001, c = 0.5;                                                            \
        for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;                                             \
        for (int it = 0; it < iters; ++it) {                                                            \
            _Pragma("unroll") for (int r = 0; r < 8; ++r) {                                             \
                REP8(INSTR)                                                                             \
            }                                                                                           \
        }                                                                                               \
        double s = 0; for (int i = 0; i < 8; ++i) s += a[i];                                            \
        if (s == 12345.678) out[threadIdx.x] = s;                                                       \
    }

#define I_FMA(k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define I_MUL(k) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define I_ADD(k) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[k]) : "v"(c));
#define I_FLOOR(k) asm volatile("v_floor_f64 %0, %0" : "+v"(a[k]));
#define I_FRACT(k) asm volatile("v_fract_f64 %0, %0" : "+v"(a[k]));
#define I_RCP(k) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[k]));
#define I_MAX(k) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[k]) : "v"(c));
#define I_CMP(k) asm volatile("v_cmp_lt_f64 vcc, %0, %1" ::"v"(a[k]), "v"(c) : "vcc");
DEF_KERNEL_D2D(k_fma, I_FMA)
DEF_KERNEL_D2D(k_mul, I_MUL)
DEF_KERNEL_D2D(k_add, I_ADD)
DEF_KERNEL_D2D(k_floor, I_FLOOR)
DEF_KERNEL_D2D(k_fract, I_FRACT)
DEF_KERNEL_D2D(k_rcp, I_RCP)
DEF_KERNEL_D2D(k_max, I_MAX)
DEF_KERNEL_D2D(k_cmp, I_CMP)

// conversions: separate int / float register sets
#define DEF_KERNEL_CVT(NAME, INSTR)                                                                     \
    __global__ void NAME(double *out, int iters) {                                                      \
        double a[8];                                                                                    \
        unsigned u[8];                                                                                  \
        float f[8];                                                                                     \
        for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x + i + 0.25; u[i] = threadIdx.x + i; f[i] = i; } \
        for (int it = 0; it < iters; ++it) {                                                            \
            _Pragma("unroll") for (int r = 0; r < 8; ++r) {                                             \
                REP8(INSTR)                                                                             \
            }                                                                                           \
        }                                                                                               \
        double s = 0; for (int i = 0; i < 8; ++i) s += a[i] + u[i] + f[i];                              \
        if (s == 12345.678) out[threadIdx.x] = s;                                                       \
    }
#define I_CVT_F64_U32(k) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(a[k]) : "v"(u[k]));
#define I_CVT_F64_I32(k) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a[k]) : "v"(u[k]));
#define I_CVT_I32_F64(k) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(u[k]) : "v"(a[k]));
#define I_CVT_U32_F64(k) asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(u[k]) : "v"(a[k]));
#define I_CVT_F64_F32(k) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[k]) : "v"(f[k]));
#define I_CVT_F32_F64(k) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[k]) : "v"(a[k]));
#define I_CVT_F32_UB0(k) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(f[k]) : "v"(u[k]));
#define I_MOV32(k) asm volatile("v_mov_b32 %0, %1" : "=v"(u[k]) : "v"(u[(k + 1) & 7]));
#define I_AND32(k) asm volatile("v_and_b32 %0, 255, %1" : "=v"(u[k]) : "v"(u[(k + 1) & 7]));
#define I_BFE(k) asm volatile("v_bfe_u32 %0, %1, 8, 8" : "=v"(u[k]) : "v"(u[(k + 1) & 7]));
#define I_MULLO(k) asm volatile("v_mul_lo_u32 %0, %1, %1" : "=v"(u[k]) : "v"(u[(k + 1) & 7]));
#define I_MAD24(k) asm volatile("v_mad_u32_u24 %0, %1, %1, %1" : "=v"(u[k]) : "v"(u[(k + 1) & 7]));
#define I_FMA32(k) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(f[k]) : "v"(f[(k + 1) & 7]));
#define I_FMAC32(k) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(f[k]) : "v"(f[(k + 1) & 7]));
#define I_MUL32(k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[k]) : "v"(f[(k + 1) & 7]));
#define I_ADD32(k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[k]) : "v"(f[(k + 1) & 7]));
#define I_RCP32(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[k]));
#define I_SQRT32(k) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[k]));
#define I_CNDMASK(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[k]) : "v"(u[(k + 1) & 7]) : "vcc");
#define I_ADDU32(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 7]));
#define I_LSHLADD(k) asm volatile("v_lshl_add_u32 %0, %1, 1, %0" : "+v"(u[k]) : "v"(u[(k + 1) & 7]));
#define I_MED3(k) asm volatile("v_med3_i32 %0, %0, 0, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 7]));
#define I_DPP(k) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(u[k]) : "v"(u[(k + 1) & 7]));
#define I_READLANE(k) asm volatile("v_readlane_b32 s20, %0, 3" ::"v"(u[k]) : "s20");
#define I_SQRT64(k) asm volatile("v_sqrt_f64 %0, %0" : "+v"(a[k]));
#define I_RSQ64(k) asm volatile("v_rsq_f64 %0, %0" : "+v"(a[k]));
#define I_LDEXP64(k) asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(a[k]));
DEF_KERNEL_CVT(k_fma_f32, I_FMA32)
DEF_KERNEL_CVT(k_fmac_f32, I_FMAC32)
DEF_KERNEL_CVT(k_mul_f32, I_MUL32)
DEF_KERNEL_CVT(k_add_f32, I_ADD32)
DEF_KERNEL_CVT(k_rcp_f32, I_RCP32)
DEF_KERNEL_CVT(k_sqrt_f32, I_SQRT32)
DEF_KERNEL_CVT(k_cndmask, I_CNDMASK)
DEF_KERNEL_CVT(k_add_u32, I_ADDU32)
DEF_KERNEL_CVT(k_lshl_add, I_LSHLADD)
DEF_KERNEL_CVT(k_med3_i32, I_MED3)
DEF_KERNEL_CVT(k_mov_dpp, I_DPP)
DEF_KERNEL_CVT(k_readlane, I_READLANE)
DEF_KERNEL_CVT(k_sqrt_f64, I_SQRT64)
DEF_KERNEL_CVT(k_rsq_f64, I_RSQ64)
DEF_KERNEL_CVT(k_ldexp_f64, I_LDEXP64)
DEF_KERNEL_CVT(k_cvt_f64_u32, I_CVT_F64_U32)
DEF_KERNEL_CVT(k_cvt_f64_i32, I_CVT_F64_I32)
DEF_KERNEL_CVT(k_cvt_i32_f64, I_CVT_I32_F64)
DEF_KERNEL_CVT(k_cvt_u32_f64, I_CVT_U32_F64)
DEF_KERNEL_CVT(k_cvt_f64_f32, I_CVT_F64_F32)
DEF_KERNEL_CVT(k_cvt_f32_f64, I_CVT_F32_F64)
DEF_KERNEL_CVT(k_cvt_f32_ub0, I_CVT_F32_UB0)
DEF_KERNEL_CVT(k_mov32, I_MOV32)
DEF_KERNEL_CVT(k_and32, I_AND32)
DEF_KERNEL_CVT(k_bfe, I_BFE)
DEF_KERNEL_CVT(k_mullo, I_MULLO)
DEF_KERNEL_CVT(k_mad24, I_MAD24)

int main() {
    double *out;
    hipMalloc(&out, 4096);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, waves_per_simd = 8, iters = 2000;
    const double clock_ghz = p.clockRate * 1e-6;      // kHz -> GHz
    printf("%d CUs, reported clock %.2f GHz; %d waves per SIMD; instructions per wave: %d\n", cus, clock_ghz, waves_per_simd, iters * 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
#define RUN(K)                                                                                               \
    do {                                                                                                     \
        hipLaunchKernelGGL(K, dim3(cus * 4 * waves_per_simd / 4), dim3(256), 0, 0, out, 10);                 \
        hipDeviceSynchronize();                                                                              \
        hipEventRecord(e0);                                                                                  \
        hipLaunchKernelGGL(K, dim3(cus * 4 * waves_per_simd / 4), dim3(256), 0, 0, out, iters);              \
        hipEventRecord(e1);                                                                                  \
        hipEventSynchronize(e1);                                                                             \
        float ms;                                                                                            \
        hipEventElapsedTime(&ms, e0, e1);                                                                    \
        const double inst_per_simd = (double)waves_per_simd * iters * 64;                                    \
        printf("%-16s %8.3f ms  -> %.2f cycles per wave-instruction per SIMD (at %.2f GHz)\n", #K, ms,        \
               ms * 1e-3 * clock_ghz * 1e9 / inst_per_simd, clock_ghz);                                      \
    } while (0)
    RUN(k_fma); RUN(k_mul); RUN(k_add); RUN(k_floor); RUN(k_fract); RUN(k_rcp); RUN(k_max); RUN(k_cmp);
    RUN(k_cvt_f64_u32); RUN(k_cvt_f64_i32); RUN(k_cvt_i32_f64); RUN(k_cvt_u32_f64); RUN(k_cvt_f64_f32); RUN(k_cvt_f32_f64); RUN(k_cvt_f32_ub0);
    RUN(k_mov32); RUN(k_and32); RUN(k_bfe); RUN(k_mullo); RUN(k_mad24);
    RUN(k_fma_f32); RUN(k_fmac_f32); RUN(k_mul_f32); RUN(k_add_f32); RUN(k_rcp_f32); RUN(k_sqrt_f32); RUN(k_cndmask); RUN(k_add_u32);
    RUN(k_lshl_add); RUN(k_med3_i32); RUN(k_mov_dpp); RUN(k_readlane); RUN(k_sqrt_f64); RUN(k_rsq_f64); RUN(k_ldexp_f64);
    return 0;
}
