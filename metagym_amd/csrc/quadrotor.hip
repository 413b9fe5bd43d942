// quadrotor.hip — batched Quadrotor engine for gfx950: one lane per environment.
//
// Replaces, for N environments per launch, the reference's per-object Python hot path
//   QuadrotorSim._run_internal   metagym/quadrotor/quadrotorsim.py:122-210  (substep())
//   QuadrotorSim._check_failure  quadrotorsim.py:212-221                    (fail test in substep loop)
//   QuadrotorSim.step            quadrotorsim.py:295-304                    (sub-step loop)
//   get_sensor / get_state / _get_pitch_roll_yaw  quadrotorsim.py:111-120,260-293  (observe())
//   Quadrotor.step / _get_reward / _check_collision  env.py:127-165,211-260 (finish_step())
//   Quadrotor.reset / QuadrotorSim.reset  env.py:116-125, quadrotorsim.py:239-258 (reset kernel)
//
// Design (MI355X): the state of one env is 116 B of SoA (pos f32x3, vel f64x3, omega f64x3,
// propw f32x4, R f32x9, ct i32). A lane loads its env's state with 23 coalesced loads (lane e reads
// base[c*N + e]), runs all int(dt/precision)=10 Euler sub-steps in registers, and writes state,
// obs, reward and done back once. The 16-float observation row is transposed through LDS so the
// wave stores 4 KiB contiguous as dwordx4. No MFMA: the work is ~4 kflop of small-vector f32/f64
// algebra per env-step with no contraction dimension to tile.
//
// Precision: the reference mixes f32 arrays, f64 arrays and python floats; the dtype NumPy (NEP 50)
// evaluates each expression in is mirrored exactly (see oracle/quadrotor_oracle.c for the annotated
// restatement this kernel is tested against). The translation unit is compiled with
// -ffp-contract=off so no a*b+c is fused — NumPy never fuses — which makes this kernel bit-identical
// to the CPU oracle for everything except atan2f (OCML vs glibc, <= 2 ulp).
#include "mg_common.h"
#include "mg_philox.h"

namespace {

#ifndef MG_QUAD_BLOCK
#define MG_QUAD_BLOCK 256     // launch-shape experiments (scripts/quad_variants.py; profiles/r05/quad_launch_shapes.txt): 64 / 128 / 256
#endif
constexpr int BLOCK = MG_QUAD_BLOCK;
constexpr int WAVES_PER_BLOCK = BLOCK / mg::WAVE;
constexpr int OBS_DIM = 16;

// Derived constants: everything the reference computes from python floats before touching an
// array is folded on the host, in the same double arithmetic, then "weak"-cast where NumPy would.
struct QuadK {
    // weak python floats that meet f32 operands first (quadrotorsim.py:136-145,154-156)
    float phi32, phi_over_ra32, inv_jm32, mm32, prec32, ct0_32, ct1_32;
    float quality32, dt32, zoff32, healthy32, fail_range_sq32;
    float lm[4];        // ||prop_coord[i]||, quadrotorsim.py:146
    float pc[12];       // prop_coord
    float iinv[9];      // inverse inertia (f32), quadrotorsim.py:64
    float df[9], dm[9]; // drag matrices
    float cog[3];
    // doubles used against f64 operands
    double prec, half_dt2, half_dt, ct2, quality, inv_quality;
    double min_v, max_v, fail_velocity, fail_w, healthy, xoff, yoff;
    int quality_recip_exact;  // 1/quality is a power of two -> x / quality == x * inv_quality bit-for-bit
    int times, nt, task;
    // fused auto-reset (not in the reference: replaces the user's `if done: env.reset()` round trip)
    int auto_reset;
    float init_v_base[3], init_w_base[3];   // cfg['init_velocity'] / ['init_angular_velocity'] x,y,z (f32 arrays)
    double init_v_noisy, init_w_noisy;      // ... ['noisy']
    uint64_t seed, env_id_base;
    const int32_t *map;
    int map_h, map_w;
    const float *vtargets;   // velocity_control target trajectory [nt][3]
    int obs_dim;             // 16, or 19 for velocity_control
};

struct Lane {       // one environment, in registers
    float p[3];
    double v[3];
    double w[3];
    float pw[4];
    float R[9];
    double Rd[9];   // R widened to f64, exactly (double)R[i]: inv3 produces it, the next sub-step's R @ a reuses it
    float Ri[9];    // inv(R): _coordination_converter_to_body
    double nv, nw;  // ||v||, ||w|| of the current state (shared by drag and the failure test)
    float power;
};

// ---- f32 / f64 3x3 helpers -------------------------------------------------------------------------
// NumPy's elementwise ops never fuse, but np.matmul / np.linalg.norm run OpenBLAS kernels that do.
// The associations below are the ones that reproduce NumPy bit-for-bit (oracle/quadrotor_oracle.c
// documents the probe); the file is compiled with -ffp-contract=off so only these explicit FMAs fuse.

__device__ __forceinline__ double dot_row_f32f64(const float *row, const double *x) {
    // np.matmul(f32[3,3], f64[3]): matrix widened, dgemv association fma(M2,x2, fma(M0,x0, M1*x1))
    return fma((double)row[2], x[2], fma((double)row[0], x[0], (double)row[1] * x[1]));
}

__device__ __forceinline__ void mv_f32f64(const float *M, const double *x, double *y) {
#pragma unroll
    for (int r = 0; r < 3; ++r) y[r] = dot_row_f32f64(&M[3 * r], x);
}

__device__ __forceinline__ void mv_f32(const float *M, const float *x, float *y) {
    // np.matmul(f32[3,3], f32[3]) -> OpenBLAS sgemv: rows 0 and 1 (a SIMD pair) are the plain sum without
    // FMA, row 2 (scalar tail) uses the dgemv-style association (oracle/quadrotor_oracle.c documents the probe)
#pragma unroll
    for (int r = 0; r < 2; ++r) y[r] = (M[3 * r] * x[0] + M[3 * r + 1] * x[1]) + M[3 * r + 2] * x[2];
    y[2] = fmaf(M[8], x[2], fmaf(M[6], x[0], M[7] * x[1]));
}

__device__ __forceinline__ void mm_f32(const float *A, const float *B, float *C) {
    // sgemm association: left-to-right FMA chain over k
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            C[3 * r + c] = fmaf(A[3 * r + 2], B[6 + c], fmaf(A[3 * r + 1], B[3 + c], A[3 * r] * B[c]));
}

__device__ __forceinline__ void cross_f32(const float *a, const float *b, float *c) {
    // numpy.cross: every product rounded, then subtracted
    float t0 = a[1] * b[2], t1 = a[2] * b[1];
    float t2 = a[2] * b[0], t3 = a[0] * b[2];
    float t4 = a[0] * b[1], t5 = a[1] * b[0];
    c[0] = t0 - t1;
    c[1] = t2 - t3;
    c[2] = t4 - t5;
}

__device__ __forceinline__ double norm3(const double *x) {
    return sqrt(fma(x[2], x[2], fma(x[1], x[1], x[0] * x[0])));
}
// np.linalg.norm(f32[3])^2 = OpenBLAS sdot(x, x): every product rounded to float32, the three products
// accumulated in double, the sum rounded to float32
__device__ __forceinline__ float sumsq3(const float *x) {
    const float p0 = x[0] * x[0], p1 = x[1] * x[1], p2 = x[2] * x[2];
    return (float)(((double)p0 + (double)p1) + (double)p2);
}

// np.linalg.inv on a float32 matrix (quadrotorsim.py:207): numpy promotes to float64, solves, and
// casts back, i.e. it returns the correctly rounded f32 inverse. Same here: adjugate / det in f64
// (branch-free, ~60 f64 ops, one division), rounded to f32. R drifts away from orthonormal (the
// reference never re-normalises it), so R^T is NOT a substitute.
__device__ __forceinline__ void inv3(const float *Af, float *Ainv, double *A) {   // A: out, the widened input
#pragma unroll
    for (int i = 0; i < 9; ++i) A[i] = (double)Af[i];
    // a*b - c*d as fma(a, b, -(c*d)): one rounding less per cofactor and one instruction less; this is OUR
    // way of reaching the correctly rounded float32 inverse, not an operation of the reference, so fusing
    // is free as long as oracle and kernel do the same
    const double c00 = fma(A[4], A[8], -(A[5] * A[7]));
    const double c01 = fma(A[5], A[6], -(A[3] * A[8]));
    const double c02 = fma(A[3], A[7], -(A[4] * A[6]));
    const double det = fma(A[2], c02, fma(A[1], c01, A[0] * c00));
    const double r = 1.0 / det;
    Ainv[0] = (float)(c00 * r);
    Ainv[3] = (float)(c01 * r);
    Ainv[6] = (float)(c02 * r);
    Ainv[1] = (float)(fma(A[2], A[7], -(A[1] * A[8])) * r);
    Ainv[4] = (float)(fma(A[0], A[8], -(A[2] * A[6])) * r);
    Ainv[7] = (float)(fma(A[1], A[6], -(A[0] * A[7])) * r);
    Ainv[2] = (float)(fma(A[1], A[5], -(A[2] * A[4])) * r);
    Ainv[5] = (float)(fma(A[2], A[3], -(A[0] * A[5])) * r);
    Ainv[8] = (float)(fma(A[0], A[4], -(A[1] * A[3])) * r);
}

// ---- one 1 ms Euler sub-step, quadrotorsim.py:122-210 ------------------------------------------
// eff32[i]: the clamped voltage already rounded to f32 (quadrotorsim.py:130-134 + weak cast).
// SIMPLE = the structure of the stock config.json: diagonal drag / inertia matrices, zero centre of
// gravity offset, CT[2] == 0, propellers in the z = 0 plane. Multiplying by those structural zeros
// only ever adds +-0 to a finite sum, so the SIMPLE path is bit-identical to the general one while
// needing ~40 fewer scalar constants and ~70 fewer VALU ops per sub-step.
// want_power: self.power (:139,:188) is overwritten by every sub-step and read only by the reward
// after the last one (env.py:217), so the four f32 divisions behind it run in the last sub-step only.
template <bool SIMPLE>
__device__ __forceinline__ void substep(const QuadK &k, Lane &s, const float *eff32, bool want_power) {
    float prop_force_z = 0.0f;
    float prop_torque[3] = {0.0f, 0.0f, 0.0f};
    float me[4], pp[4];

    // :147-148 body_velocity = Rinv @ v is identical for all four propellers; only [2] is used
    const double bvz = dot_row_f32f64(&s.Ri[6], s.v);

#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float e32 = eff32[i];
        float phi_w = k.phi32 * s.pw[i];                         // :136
        me[i] = k.phi_over_ra32 * (e32 - phi_w);                 // :137-138
        if (want_power) pp[i] = fabsf(me[i] / k.phi32 * e32);    // :139
        float d_prop_w = k.inv_jm32 * (me[i] - k.mm32);          // :141-142
        float w_m = s.pw[i] + k.prec32 * d_prop_w;               // :144-145
        const float *pc = &k.pc[3 * i];
        // :149-151 (omega x coord)[2] * l_m, f64
        double cz = s.w[0] * (double)pc[1] - s.w[1] * (double)pc[0];
        double v_1 = bvz + cz * (double)k.lm[i];
        float t0 = (k.ct0_32 * w_m) * w_m;                       // :154 f32 chain
        double t1 = (double)(k.ct1_32 * w_m) * v_1;              // :155 f32 product, widened
        double thrust = (double)t0 + t1;
        if (!SIMPLE) {
            double sign = v_1 > 0 ? 1.0 : -1.0;                  // :152
            thrust = thrust + ((k.ct2 * v_1) * v_1) * sign;      // :156 f64
        }
        s.pw[i] = w_m;                                           // :158
        prop_force_z = (float)((double)prop_force_z + thrust);   // :159 f64 add, f32 store
        const float T = (float)thrust;                           // :160-162 cross(-[0,0,T], coord)
        if (SIMPLE) {
            prop_torque[0] += T * pc[1];
            prop_torque[1] += (-T) * pc[0];
        } else {
            float a[3] = {-0.0f, -0.0f, -T};
            float cr[3];
            cross_f32(a, pc, cr);
            prop_torque[0] += cr[0];
            prop_torque[1] += cr[1];
            prop_torque[2] += cr[2];
        }
    }
    prop_torque[2] += ((-me[0] + me[1]) - me[2]) + me[3];        // :164

    // :166-172 drag: -||v|| * ((Df @ Rinv) @ v), -||w|| * (Dm @ w)
    float DfRi[9];
    double tmp[3], f_drag[3], t_drag[3];
    if (SIMPLE) {
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) DfRi[3 * r + c] = k.df[4 * r] * s.Ri[3 * r + c];
    } else {
        mm_f32(k.df, s.Ri, DfRi);
    }
    mv_f32f64(DfRi, s.v, tmp);
    const double mnv = -s.nv;
#pragma unroll
    for (int c = 0; c < 3; ++c) f_drag[c] = mnv * tmp[c];
    if (SIMPLE) {
#pragma unroll
        for (int c = 0; c < 3; ++c) tmp[c] = (double)k.dm[4 * c] * s.w[c];
    } else {
        mv_f32f64(k.dm, s.w, tmp);
    }
    const double mnw = -s.nw;
#pragma unroll
    for (int c = 0; c < 3; ++c) t_drag[c] = mnw * tmp[c];

    // :174-178 gravity, f32: Rinv @ [0,0,-9.8] * quality
    float f_grav[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) f_grav[c] = (s.Ri[3 * c + 2] * -9.8f) * k.quality32;

    // :180-184
    double t_all[3], body_acc[3], acc[3];
    float t_grav_neg[3] = {0.0f, 0.0f, 0.0f};
    if (!SIMPLE) {
        float t_grav[3];
        cross_f32(f_grav, k.cog, t_grav);
#pragma unroll
        for (int c = 0; c < 3; ++c) t_grav_neg[c] = -t_grav[c];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float pf = (c == 2) ? prop_force_z : 0.0f;
        double f_all = (double)(pf + f_grav[c]) + f_drag[c];
        t_all[c] = (double)(SIMPLE ? prop_torque[c] : prop_torque[c] + t_grav_neg[c]) + t_drag[c];
        body_acc[c] = k.quality_recip_exact ? f_all * k.inv_quality : f_all / k.quality;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)   // mv_f32f64(s.R, body_acc, acc) on the already widened matrix
        acc[r] = fma(s.Rd[3 * r + 2], body_acc[2], fma(s.Rd[3 * r], body_acc[0], s.Rd[3 * r + 1] * body_acc[1]));
    // :185-188
#pragma unroll
    for (int c = 0; c < 3; ++c)
        s.p[c] = (float)((double)s.p[c] + (s.v[c] * k.prec + k.half_dt2 * acc[c]));
#pragma unroll
    for (int c = 0; c < 3; ++c) s.v[c] = s.v[c] + k.prec * acc[c];
    if (want_power) s.power = ((pp[0] + pp[1]) + pp[2]) + pp[3];

    // :190-204 attitude
    double alpha[3], tw[3];
    if (SIMPLE) {
#pragma unroll
        for (int c = 0; c < 3; ++c) alpha[c] = (double)k.iinv[4 * c] * t_all[c];
    } else {
        mv_f32f64(k.iinv, t_all, alpha);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) tw[c] = s.w[c] + k.half_dt * alpha[c];
    // skew(tw) rounded to f32 (:193-199); R += dt * (R @ S) with the sgemm FMA association, the
    // structural zero of each S column folded away (0*x contributes +-0)
    const float s01 = (float)(-tw[2]), s02 = (float)(tw[1]);
    const float s10 = (float)(tw[2]), s12 = (float)(-tw[0]);
    const float s20 = (float)(-tw[1]), s21 = (float)(tw[0]);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float r0 = s.R[3 * r], r1 = s.R[3 * r + 1], r2 = s.R[3 * r + 2];
        const float rs0 = fmaf(r2, s20, fmaf(r1, s10, r0 * 0.0f));
        const float rs1 = fmaf(r2, s21, fmaf(r1, 0.0f, r0 * s01));
        const float rs2 = fmaf(r2, 0.0f, fmaf(r1, s12, r0 * s02));
        s.R[3 * r] = r0 + k.prec32 * rs0;
        s.R[3 * r + 1] = r1 + k.prec32 * rs1;
        s.R[3 * r + 2] = r2 + k.prec32 * rs2;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) s.w[c] = s.w[c] + k.prec * alpha[c];
    inv3(s.R, s.Ri, s.Rd);                                       // :206-208
    s.nv = norm3(s.v);
    s.nw = norm3(s.w);
}

// quadrotorsim.py:212-221
__device__ __forceinline__ int failure_code(const QuadK &k, const Lane &s) {
    if (sumsq3(s.p) > k.fail_range_sq32) return 1;   // == sqrtf(sumsq) > fail_range32, see fold_config
    if (s.nv > k.fail_velocity) return 2;
    if (s.nw > k.fail_w) return 3;
    return 0;
}

// quadrotorsim.py:260-293, :111-120; env.py:193-209
__device__ __forceinline__ void observe(const QuadK &k, const Lane &s, float *obs) {
    double b_v[3];
    float b_pos[3], imu[3];
    const float g[3] = {0.0f, 0.0f, -9.8f};
    mv_f32f64(s.Ri, s.v, b_v);
    mv_f32(s.Ri, s.p, b_pos);
    mv_f32(s.Ri, g, imu);
    float roll = atan2f(s.R[7], s.R[8]);
    float pitch = atan2f(-s.R[6], sqrtf(s.R[7] * s.R[7] + s.R[8] * s.R[8]));
    float yaw = atan2f(s.R[3], s.R[0]);
    obs[0] = (float)b_v[0]; obs[1] = (float)b_v[1]; obs[2] = (float)b_v[2];
    obs[3] = b_pos[0]; obs[4] = b_pos[1]; obs[5] = b_pos[2];
    obs[6] = 0.0f + imu[0]; obs[7] = 0.0f + imu[1]; obs[8] = 0.0f + imu[2];
    obs[9] = (float)s.w[0]; obs[10] = (float)s.w[1]; obs[11] = (float)s.w[2];
    obs[12] = pitch; obs[13] = roll; obs[14] = yaw;
    obs[15] = s.p[2] + k.zoff32;
}

// python slice normalisation map[a:b] for a dimension of length len
__device__ __forceinline__ void py_slice(long a, long b, long len, long &lo, long &hi) {
    if (a < 0) { a += len; if (a < 0) a = 0; } else if (a > len) a = len;
    if (b < 0) { b += len; if (b < 0) b = 0; } else if (b > len) b = len;
    lo = a; hi = b;
}

// env.py:248-260
__device__ __forceinline__ bool collision(const QuadK &k, const double *op, const double *np_) {
    long mn[3], mx[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double lo = op[i] < np_[i] ? op[i] : np_[i];
        double hi = op[i] > np_[i] ? op[i] : np_[i];
        mn[i] = (long)floor(lo);
        mx[i] = (long)ceil(hi);
    }
    int any = 0;
    if (k.map != nullptr) {
        long y0, y1, x0, x1;
        py_slice(mn[1], mx[1] + 1, k.map_h, y0, y1);
        py_slice(mn[0], mx[0] + 1, k.map_w, x0, x1);
        for (long y = y0; y < y1; ++y)
            for (long x = x0; x < x1; ++x) any |= (k.map[y * k.map_w + x] != 0);
    }
    return (mn[2] < any) || (mx[2] < any);  // heights compared with the *bool* np.any(...)
}

// ---- pre-reset, all-float32 sub-step: define_velocity_control_task, quadrotorsim.py:306-319 -------
// Before reset() every simulator array is float32 (quadrotorsim.py:20-28), so _run_internal runs
// entirely in float32 there (python floats are weak). Only used once per env object to roll the
// velocity_control target trajectory; general (non-specialised) formulation, one lane.
struct Lane32 {
    float p[3], v[3], w[3], pw[4], R[9], Ri[9];
};

__device__ void substep_f32state(const QuadK &k, Lane32 &s, const float *eff32) {
    float prop_force_z = 0.0f, prop_torque[3] = {0.0f, 0.0f, 0.0f}, me[4];
    const float ct2_32 = (float)k.ct2;
    for (int i = 0; i < 4; ++i) {
        const float e32 = eff32[i];
        float phi_w = k.phi32 * s.pw[i];
        me[i] = k.phi_over_ra32 * (e32 - phi_w);
        float d_prop_w = k.inv_jm32 * (me[i] - k.mm32);
        float w_m = s.pw[i] + k.prec32 * d_prop_w;
        const float *pc = &k.pc[3 * i];
        float bv[3], cr[3];
        mv_f32(s.Ri, s.v, bv);
        cross_f32(s.w, pc, cr);
        float v_1 = bv[2] + cr[2] * k.lm[i];
        float sign = v_1 > 0 ? 1.0f : -1.0f;
        float thrust = ((k.ct0_32 * w_m) * w_m + (k.ct1_32 * w_m) * v_1) + ((ct2_32 * v_1) * v_1) * sign;
        s.pw[i] = w_m;
        prop_force_z = prop_force_z + thrust;
        float a[3] = {-0.0f, -0.0f, -thrust};
        cross_f32(a, pc, cr);
        prop_torque[0] += cr[0]; prop_torque[1] += cr[1]; prop_torque[2] += cr[2];
    }
    prop_torque[2] += ((-me[0] + me[1]) - me[2]) + me[3];
    float DfRi[9], tmp[3], f_drag[3], t_drag[3];
    mm_f32(k.df, s.Ri, DfRi);
    mv_f32(DfRi, s.v, tmp);
    const float nv = -sqrtf(sumsq3(s.v));
    for (int c = 0; c < 3; ++c) f_drag[c] = nv * tmp[c];
    mv_f32(k.dm, s.w, tmp);
    const float nw = -sqrtf(sumsq3(s.w));
    for (int c = 0; c < 3; ++c) t_drag[c] = nw * tmp[c];
    const float g[3] = {0.0f, 0.0f, -9.8f};
    float f_grav[3], t_grav[3], body_acc[3], t_all[3], acc[3];
    mv_f32(s.Ri, g, f_grav);
    for (int c = 0; c < 3; ++c) f_grav[c] = f_grav[c] * k.quality32;
    cross_f32(f_grav, k.cog, t_grav);
    for (int c = 0; c < 3; ++c) {
        const float pf = (c == 2) ? prop_force_z : 0.0f;
        const float f_all = (pf + f_grav[c]) + f_drag[c];
        t_all[c] = (prop_torque[c] + (-t_grav[c])) + t_drag[c];
        body_acc[c] = f_all / k.quality32;
    }
    mv_f32(s.R, body_acc, acc);
    const float half_dt2 = (float)k.half_dt2, half_dt = (float)k.half_dt;
    for (int c = 0; c < 3; ++c) s.p[c] = s.p[c] + (s.v[c] * k.prec32 + half_dt2 * acc[c]);
    for (int c = 0; c < 3; ++c) s.v[c] = s.v[c] + k.prec32 * acc[c];
    float alpha[3], tw[3];
    mv_f32(k.iinv, t_all, alpha);
    for (int c = 0; c < 3; ++c) tw[c] = s.w[c] + half_dt * alpha[c];
    float S[9] = {0.0f, -tw[2], tw[1], tw[2], 0.0f, -tw[0], -tw[1], tw[0], 0.0f};
    float RS[9];
    mm_f32(s.R, S, RS);
    for (int c = 0; c < 9; ++c) s.R[c] = s.R[c] + k.prec32 * RS[c];
    for (int c = 0; c < 3; ++c) s.w[c] = s.w[c] + k.prec32 * alpha[c];
    double Rd[9];
    inv3(s.R, s.Ri, Rd);
}

__global__ void quadrotor_targets_kernel(QuadK k, int nt, const float *actions, float *targets) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    Lane32 s;
    for (int c = 0; c < 3; ++c) { s.p[c] = 0.0f; s.v[c] = 0.0f; s.w[c] = 0.0f; }
    for (int c = 0; c < 4; ++c) s.pw[c] = 0.0f;
    for (int c = 0; c < 9; ++c) { s.R[c] = (c % 4 == 0) ? 1.0f : 0.0f; s.Ri[c] = s.R[c]; }
    for (int t = 0; t < nt; ++t) {
        float eff32[4];
        for (int i = 0; i < 4; ++i) {
            double d = (double)actions[4 * t + i];
            d = d > k.max_v ? k.max_v : (d < k.min_v ? k.min_v : d);
            eff32[i] = (float)d;
        }
        for (int it = 0; it < k.times; ++it) substep_f32state(k, s, eff32);
        targets[3 * t] = s.v[0]; targets[3 * t + 1] = s.v[1]; targets[3 * t + 2] = s.v[2];
    }
}

// ---- SoA load / store ----------------------------------------------------------------------------

__device__ __forceinline__ void load_lane(const mg_quadrotor_state &st, int n, int e, Lane &s, int &ct) {
#pragma unroll
    for (int c = 0; c < 3; ++c) s.p[c] = st.pos[(size_t)c * n + e];
#pragma unroll
    for (int c = 0; c < 3; ++c) s.v[c] = st.vel[(size_t)c * n + e];
#pragma unroll
    for (int c = 0; c < 3; ++c) s.w[c] = st.omega[(size_t)c * n + e];
#pragma unroll
    for (int c = 0; c < 4; ++c) s.pw[c] = st.propw[(size_t)c * n + e];
#pragma unroll
    for (int c = 0; c < 9; ++c) s.R[c] = st.rot[(size_t)c * n + e];
    ct = st.ct[e];
    inv3(s.R, s.Ri, s.Rd);
    s.nv = norm3(s.v);
    s.nw = norm3(s.w);
    s.power = 0.0f;
}

// Every output of a step is written once and next read by a later launch (or by the caller), never by
// this one: streaming (nontemporal) stores let the bytes leave through the fabric as they are issued
// instead of waiting in L2 for the end-of-kernel write-back (-2..3 % at 65 536 envs).
template <typename T> __device__ __forceinline__ void st_stream(T *p, T v) { __builtin_nontemporal_store(v, p); }
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void store_lane(const mg_quadrotor_state &st, int n, int e, const Lane &s, int ct) {
#pragma unroll
    for (int c = 0; c < 3; ++c) st_stream(&st.pos[(size_t)c * n + e], s.p[c]);
#pragma unroll
    for (int c = 0; c < 3; ++c) st_stream(&st.vel[(size_t)c * n + e], s.v[c]);
#pragma unroll
    for (int c = 0; c < 3; ++c) st_stream(&st.omega[(size_t)c * n + e], s.w[c]);
#pragma unroll
    for (int c = 0; c < 4; ++c) st_stream(&st.propw[(size_t)c * n + e], s.pw[c]);
#pragma unroll
    for (int c = 0; c < 9; ++c) st_stream(&st.rot[(size_t)c * n + e], s.R[c]);
    st_stream(&st.ct[e], ct);
}

// Transpose the wave's 64 x 16 observation rows through LDS and store them as 4 coalesced
// dwordx4 sweeps (each wave instruction writes 1 KiB contiguous). Rows are padded to 17 floats so
// the per-lane row writes hit distinct banks; a partial last wave falls back to per-row stores.
__device__ __forceinline__ void store_obs_wave(float *tile, const float *obs, float *out, int n, int e, int obs_dim) {
    if (obs_dim != OBS_DIM) {   // velocity_control rows (19 floats) are not 16-byte aligned: plain row stores
        if (e < n)
            for (int c = 0; c < obs_dim; ++c) out[(size_t)e * obs_dim + c] = obs[c];
        return;
    }
    const int lane = threadIdx.x & (mg::WAVE - 1);
    const int wave_base = e - lane;                 // first env of this wave
    const bool full = (wave_base + mg::WAVE) <= n;  // wave-uniform
    if (full) {
#pragma unroll
        for (int c = 0; c < OBS_DIM; ++c) tile[lane * (OBS_DIM + 1) + c] = obs[c];
        __builtin_amdgcn_wave_barrier();
        float4 *dst = reinterpret_cast<float4 *>(out + (size_t)wave_base * OBS_DIM);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = j * mg::WAVE + lane;      // float4 index inside the wave's 4 KiB block
            const int row = q >> 2, col = (q & 3) * 4;
            const float *src = &tile[row * (OBS_DIM + 1) + col];
            st_stream(reinterpret_cast<v4f *>(dst + q), v4f{src[0], src[1], src[2], src[3]});
        }
        __builtin_amdgcn_wave_barrier();
    } else if (e < n) {
        float4 *dst = reinterpret_cast<float4 *>(out + (size_t)e * OBS_DIM);
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[j] = make_float4(obs[4 * j], obs[4 * j + 1], obs[4 * j + 2], obs[4 * j + 3]);
    }
}

// QuadrotorSim.reset quadrotorsim.py:239-258 with the noise drawn on the device:
// value = base + noisy * U[0,1) * (+1 if U' > 0.5 else -1), per component.
__device__ __forceinline__ void reset_lane_random(const QuadK &k, Lane &s, int e, uint32_t episode) {
    // two Philox blocks = 8 words: six 32-bit magnitudes and one word of sign bits (the reset sits on the
    // critical path of every wave that holds a finished env, so a third block is worth avoiding).
    // counter = (global env id lo, hi, episode, block), key = seed: the k-th auto-reset of a given env draws
    // the same noise whatever the sharding, the launch shape (steps per launch) or the host did in between,
    // and no launch argument changes from step to step (hipGraph replay; oracle: qo_reset_random)
    uint32_t r[8];
    const uint64_t gid = k.env_id_base + (uint64_t)e;
#pragma unroll
    for (int d = 0; d < 2; ++d)
        philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), episode, (uint32_t)d, (uint32_t)k.seed,
                      (uint32_t)(k.seed >> 32), &r[4 * d]);
    const double inv32 = 1.0 / 4294967296.0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        s.p[c] = 0.0f;
        const double sv = ((r[6] >> c) & 1u) ? 1.0 : -1.0;
        const double sw = ((r[6] >> (3 + c)) & 1u) ? 1.0 : -1.0;
        s.v[c] = (double)k.init_v_base[c] + (k.init_v_noisy * ((double)r[c] * inv32)) * sv;
        s.w[c] = (double)k.init_w_base[c] + (k.init_w_noisy * ((double)r[3 + c] * inv32)) * sw;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) s.pw[c] = 0.0f;
#pragma unroll
    for (int c = 0; c < 9; ++c) {
        s.R[c] = (c % 4 == 0) ? 1.0f : 0.0f;
        s.Ri[c] = s.R[c];
        s.Rd[c] = (double)s.R[c];
    }
    s.nv = norm3(s.v);
    s.nw = norm3(s.w);
    s.power = 0.0f;
}

// ---- kernels ---------------------------------------------------------------------------------------

struct StepIO {
    const float *action;   // [T][n][4]
    float *obs;            // [T][n][16]
    float *reward;         // [T][n] or null
    double *reward64;      // [T][n] or null
    uint8_t *done;         // [T][n]
    uint8_t *failed;       // [T][n] or null
};

// The step kernel's argument block as the kernarg segment lays it out, and a pointer to it that the
// optimiser cannot relate to the kernel's own argument loads (see the epilogue of the step kernel).
struct KArgs { QuadK k; mg_quadrotor_state st; StepIO io; int n; int n_steps; };
typedef const __attribute__((address_space(4))) KArgs KArgsC;
__device__ __forceinline__ KArgsC *kernargs_fresh() {
    KArgsC *p = (KArgsC *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

template <bool SIMPLE>
__global__ __launch_bounds__(BLOCK) void quadrotor_step_kernel(QuadK k, mg_quadrotor_state st, StepIO io,
                                                               int n, int n_steps) {
    __shared__ float tiles[WAVES_PER_BLOCK][mg::WAVE * (OBS_DIM + 1)];
    const int e = blockIdx.x * BLOCK + threadIdx.x;
    const bool live = e < n;
    const int el = live ? e : n - 1;  // out-of-range lanes shadow the last env, stores are masked
    float *tile = tiles[threadIdx.x / mg::WAVE];

    // Latency, not bandwidth, bounds a 65 536-env launch (one wave per SIMD). Three round trips used to
    // run back to back: state loads, then ~9 scalar kernarg-line misses scattered through the prologue,
    // then the action load. Issue them together: touch every 64-byte line of the 560-byte kernarg
    // segment now (later s_loads hit the scalar cache) and fetch the first action before the state.
    {
        const uint32_t *ka = (const uint32_t *)__builtin_amdgcn_kernarg_segment_ptr();
        uint32_t touch = 0;
#pragma unroll
        for (int line = 0; line < (int)((sizeof(QuadK) + sizeof(mg_quadrotor_state) + sizeof(StepIO) + 8 + 63) / 64); ++line)
            touch |= ka[line * 16];
        asm volatile("" ::"s"(touch));
    }
    float4 a_next = reinterpret_cast<const float4 *>(io.action)[el];
    __builtin_amdgcn_sched_barrier(0);   // keep this load ahead of the state loads (it would be sunk to its first use)

    Lane s;
    int ct;
    uint32_t episode = 0;
    if (k.auto_reset) episode = st.episode[el];
    const uint32_t episode_in = episode;
    load_lane(st, n, el, s, ct);
    // All prologue loads land here (vmcnt = 0), not at their first use inside the step loop: there the wait
    // would be re-executed by every later step of a rollout and would also drain that step's freshly issued
    // action prefetch and the previous step's stores (2.5 us per step at K > 1; free at K = 1).
    __builtin_amdgcn_s_waitcnt(0x0F70);

    for (int t = 0; t < n_steps; ++t) {
        const size_t off = (size_t)t * n;
        const float4 a = a_next;
        if (t + 1 < n_steps)      // prefetch the next step's action behind this step's arithmetic
            a_next = reinterpret_cast<const float4 *>(io.action)[off + n + el];
        // quadrotorsim.py:130-134: clamp the (f32-valued) python float against python floats
        const float av[4] = {a.x, a.y, a.z, a.w};
        float eff32[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double d = (double)av[i];
            d = d > k.max_v ? k.max_v : (d < k.min_v ? k.min_v : d);
            eff32[i] = (float)d;
        }
        ct += 1;                                                            // env.py:128
        const double old_pos[3] = {(double)s.p[0] + k.xoff, (double)s.p[1] + k.yoff,
                                   (double)(s.p[2] + k.zoff32)};            // env.py:131-133
        int fail = 0;
        for (int it = 0; it < k.times; ++it) {                              // quadrotorsim.py:302-304
            if (fail == 0) {      // a failed env freezes at the failing sub-step (reference raises)
                substep<SIMPLE>(k, s, eff32, it == k.times - 1);
                fail = failure_code(k, s);
            }
        }
        // The reward / observation constants and the output pointers are used only from here on. Held in
        // SGPRs across the sub-step loop they overflow the scalar file and get spilled to VGPR lanes
        // (125 spills, ~200 v_writelane/v_readlane on the hot path); re-reading them from the kernarg
        // segment (scalar-cache hits, the prologue touched every line) through a pointer the compiler
        // cannot connect to the prologue's loads keeps the loop's SGPR budget for the loop.
        const KArgsC *kae = kernargs_fresh();
        const QuadK &ke = *(const QuadK *)&kae->k;
        const StepIO &ioe = *(const StepIO *)&kae->io;
        const bool vel_task = ke.task == MG_QUADROTOR_TASK_VELOCITY_CONTROL;
        // _update_state env.py:262-273: the observation's target entries come from min(ct, nt-1) with ct
        // already incremented and not yet cleared by the episode end
        const int tn_step = ct < ke.nt - 1 ? ct : ke.nt - 1;
        double reward = 0.0;
        int done = 1;
        if (fail == 0 && vel_task) {
            // env.py:150-157: body-frame target = Rinv(f32) @ target(f32); reward -0.001 * L1 difference
            float bt[3];
            mv_f32(s.Ri, &ke.vtargets[3 * (ct - 1)], bt);
            double b_v[3];
            mv_f32f64(s.Ri, s.v, b_v);
            const double diff = (fabs((double)bt[0] - b_v[0]) + fabs((double)bt[1] - b_v[1])) + fabs((double)bt[2] - b_v[2]);
            const float energy = ke.dt32 * s.power;
            const double r = (ke.healthy32 < energy) ? -ke.healthy : -(double)energy;
            reward = r + (-0.001 * diff);
            done = 0;
            if (ct == ke.nt) { done = 1; ct = 0; }
        } else if (fail == 0) {
            const double new_pos[3] = {(double)s.p[0] + ke.xoff, (double)s.p[1] + ke.yoff,
                                       (double)(s.p[2] + ke.zoff32)};
            const bool hit = collision(ke, old_pos, new_pos);                // env.py:145
            // _get_reward env.py:211-246
            const float energy = ke.dt32 * s.power;
            double r = (ke.healthy32 < energy) ? -ke.healthy : -(double)energy;   // -min(energy, healthy)
            double task_reward = hit ? 0.0 : ke.healthy;
            if (ke.task == MG_QUADROTOR_TASK_HOVERING_CONTROL) {
                task_reward -= 1.0 * s.nv + 1.0 * s.nw;
                const float z_move = fabsf(0.0f - s.p[2]);   // pos_0 is the reset position = 0 (env.py:123)
                if (z_move < 0.5f) task_reward += 10;
                else {
                    const float o = 0.5f - z_move;
                    task_reward += (o > -20.0f) ? (double)o : -20.0;
                }
            }
            if (ke.task == MG_QUADROTOR_TASK_HOVERING_CONTROL || ke.healthy32 < energy)
                reward = r + task_reward;     // np.float64 task_reward, or python floats on both sides
            else                              // no_collision: np.float32 + weak python float -> f32 add (env.py:220-221)
                reward = (double)((float)r + (float)task_reward);
            done = 0;
            if (hit) { done = 1; ct = 0; }                                  // env.py:147-149
            if (ct == ke.nt) { done = 1; ct = 0; }                           // env.py:159-161
        } else {
            ct = 0;
        }
        // Reward / done above read the stepped state only, so the observation is computed ONCE, after the
        // optional in-place reset: stepped state for running envs, and — vector-env convention — the first
        // observation of the next episode for the envs that just finished.
        int tn = tn_step;
        if (ke.auto_reset && done) {
            reset_lane_random(ke, s, el, episode);
            episode += 1;
            tn = ct < ke.nt - 1 ? ct : ke.nt - 1;
        }
        // The next step's action (requested at the top of this step) is taken out of flight here, before this
        // step's stores are issued: vmcnt counts loads and stores in one queue, so a wait placed at the top of
        // the next step would also wait for every store below.
        if (t + 1 < n_steps) asm volatile("" : "+v"(a_next.x), "+v"(a_next.y), "+v"(a_next.z), "+v"(a_next.w));
        // The state is final here. Its stores (63 % of the bytes this launch writes) go out before the
        // observation arithmetic so that they drain behind it instead of after it.
        if (t == n_steps - 1 && live) {
            const mg_quadrotor_state &ste = *(const mg_quadrotor_state *)&kae->st;
            store_lane(ste, n, e, s, ct);
            if (episode != episode_in) st_stream(&ste.episode[e], episode);   // rare: only lanes that restarted
        }
        __builtin_amdgcn_sched_barrier(0);   // pure arithmetic would otherwise be hoisted above the stores
        float obs[OBS_DIM + 3];
        observe(ke, s, obs);
        if (vel_task) { obs[16] = ke.vtargets[3 * tn]; obs[17] = ke.vtargets[3 * tn + 1]; obs[18] = ke.vtargets[3 * tn + 2]; }
        store_obs_wave(tile, obs, ioe.obs + off * ke.obs_dim, n, e, ke.obs_dim);
        if (live) {
            if (ioe.reward) st_stream(&ioe.reward[off + e], (float)reward);
            if (ioe.reward64) st_stream(&ioe.reward64[off + e], reward);
            st_stream(&ioe.done[off + e], (uint8_t)done);
            if (ioe.failed) st_stream(&ioe.failed[off + e], (uint8_t)fail);
        }
    }
}

__global__ __launch_bounds__(BLOCK) void quadrotor_reset_kernel(QuadK k, mg_quadrotor_state st,
                                                                const uint8_t *mask, const double *init_vel,
                                                                const double *init_omega, float *obs_out, int n) {
    const int e = blockIdx.x * BLOCK + threadIdx.x;
    if (e >= n) return;
    if (mask != nullptr && mask[e] == 0) return;
    Lane s;
    // _zero_state quadrotorsim.py:20-28, then the injected noise :241-254
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        s.p[c] = 0.0f;
        s.v[c] = init_vel ? init_vel[(size_t)c * n + e] : 0.0;
        s.w[c] = init_omega ? init_omega[(size_t)c * n + e] : 0.0;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) s.pw[c] = 0.0f;
#pragma unroll
    for (int c = 0; c < 9; ++c) s.R[c] = (c % 4 == 0) ? 1.0f : 0.0f;
    inv3(s.R, s.Ri, s.Rd);
    s.nv = norm3(s.v);
    s.nw = norm3(s.w);
    s.power = 0.0f;
    const int ct = st.ct[e];
    store_lane(st, n, e, s, ct);   // ct is not cleared by reset() (env.py:116-125)
    if (obs_out != nullptr) {
        float obs[OBS_DIM + 3];
        observe(k, s, obs);
        if (k.task == MG_QUADROTOR_TASK_VELOCITY_CONTROL) {
            const int tn = ct < k.nt - 1 ? ct : k.nt - 1;
            obs[16] = k.vtargets[3 * tn]; obs[17] = k.vtargets[3 * tn + 1]; obs[18] = k.vtargets[3 * tn + 2];
        }
        for (int c = 0; c < k.obs_dim; ++c) obs_out[(size_t)e * k.obs_dim + c] = obs[c];
    }
}

// ---- host: fold the config into kernel constants ----------------------------------------------

void host_inv3_f32(const float *Af, float *Ainv) {
    // np.linalg.inv(self._inertia) quadrotorsim.py:64: f64 solve, cast back to f32; same formula as the
    // device inv3 and the oracle's qo_inv3_f32 (explicit FMAs)
    double A[9];
    for (int i = 0; i < 9; ++i) A[i] = (double)Af[i];
    const double c00 = fma(A[4], A[8], -(A[5] * A[7])), c01 = fma(A[5], A[6], -(A[3] * A[8])),
                 c02 = fma(A[3], A[7], -(A[4] * A[6]));
    const double det = fma(A[2], c02, fma(A[1], c01, A[0] * c00));
    const double r = 1.0 / det;
    Ainv[0] = (float)(c00 * r); Ainv[3] = (float)(c01 * r); Ainv[6] = (float)(c02 * r);
    Ainv[1] = (float)(fma(A[2], A[7], -(A[1] * A[8])) * r);
    Ainv[4] = (float)(fma(A[0], A[8], -(A[2] * A[6])) * r);
    Ainv[7] = (float)(fma(A[1], A[6], -(A[0] * A[7])) * r);
    Ainv[2] = (float)(fma(A[1], A[5], -(A[2] * A[4])) * r);
    Ainv[5] = (float)(fma(A[2], A[3], -(A[0] * A[5])) * r);
    Ainv[8] = (float)(fma(A[0], A[4], -(A[1] * A[3])) * r);
}

int fold_config(const mg_quadrotor_config *c, QuadK *k, bool need_targets = true) {
    if (!(c->precision >= 1e-8) || c->precision > c->dt)   // quadrotorsim.py:299-300
        return mg::set_error(MG_ERR_BAD_CONFIG, "precision %g must be in [1e-8, dt=%g]", c->precision, c->dt);
    if (c->task != MG_QUADROTOR_TASK_NO_COLLISION && c->task != MG_QUADROTOR_TASK_HOVERING_CONTROL &&
        c->task != MG_QUADROTOR_TASK_VELOCITY_CONTROL)
        return mg::set_error(MG_ERR_UNSUPPORTED, "quadrotor task %d is not implemented", c->task);
    if (c->task == MG_QUADROTOR_TASK_VELOCITY_CONTROL && need_targets && c->velocity_targets_d == nullptr)
        return mg::set_error(MG_ERR_NULL_POINTER, "velocity_control needs cfg->velocity_targets_d");
    if (c->map_d != nullptr && (c->map_h <= 0 || c->map_w <= 0))
        return mg::set_error(MG_ERR_BAD_SIZE, "map shape %d x %d", c->map_h, c->map_w);
    k->phi32 = (float)c->phi;
    k->phi_over_ra32 = (float)(c->phi / c->ra);
    k->inv_jm32 = (float)(1.0 / c->jm);
    k->mm32 = (float)c->mm;
    k->prec32 = (float)c->precision;
    k->ct0_32 = (float)c->ct0;
    k->ct1_32 = (float)c->ct1;
    k->quality32 = (float)c->quality;
    k->dt32 = (float)c->dt;
    k->zoff32 = (float)c->z_offset;
    k->healthy32 = (float)c->healthy_reward;
    {
        // np.linalg.norm(pos) > fail_range, both f32 (quadrotorsim.py:213). sqrtf is monotone and
        // correctly rounded, so  sqrtf(s) > T  <=>  s > S  with S = max{ x : sqrtf(x) <= T }.
        const float T = (float)c->fail_range;
        float S = T * T;
        while (sqrtf(S) > T) S = nextafterf(S, 0.0f);
        while (sqrtf(nextafterf(S, INFINITY)) <= T) S = nextafterf(S, INFINITY);
        k->fail_range_sq32 = S;
    }
    for (int i = 0; i < 4; ++i) {
        const float *p = &c->prop_coord[3 * i];
        const float p0 = p[0] * p[0], p1 = p[1] * p[1], p2 = p[2] * p[2];      // sdot: f32 products, double sum
        k->lm[i] = sqrtf((float)(((double)p0 + (double)p1) + (double)p2));
    }
    for (int i = 0; i < 12; ++i) k->pc[i] = c->prop_coord[i];
    host_inv3_f32(c->inertia, k->iinv);
    for (int i = 0; i < 9; ++i) { k->df[i] = c->drag_f[i]; k->dm[i] = c->drag_m[i]; }
    for (int i = 0; i < 3; ++i) k->cog[i] = c->gravity_center[i];
    k->prec = c->precision;
    k->half_dt2 = 0.5 * c->precision * c->precision;   // python: 0.5 * p * p, left to right
    k->half_dt = 0.5 * c->precision;
    k->ct2 = c->ct2;
    k->quality = c->quality;
    k->inv_quality = 1.0 / c->quality;
    int ex = 0;
    (void)frexp(c->quality, &ex);
    k->quality_recip_exact = (frexp(c->quality, &ex) == 0.5) ? 1 : 0;   // power of two
    k->min_v = c->min_voltage;
    k->max_v = c->max_voltage;
    k->fail_velocity = c->fail_velocity;
    k->fail_w = c->fail_w;
    k->healthy = c->healthy_reward;
    k->xoff = (double)c->x_offset;
    k->yoff = (double)c->y_offset;
    k->times = (int)(c->dt / c->precision);            // quadrotorsim.py:302
    k->nt = c->nt;
    k->task = c->task;
    k->map = c->map_d;
    k->map_h = c->map_h;
    k->map_w = c->map_w;
    k->vtargets = c->velocity_targets_d;
    k->obs_dim = c->task == MG_QUADROTOR_TASK_VELOCITY_CONTROL ? 19 : 16;
    k->auto_reset = 0;
    for (int i = 0; i < 3; ++i) { k->init_v_base[i] = 0.0f; k->init_w_base[i] = 0.0f; }
    k->init_v_noisy = k->init_w_noisy = 0.0;
    k->seed = k->env_id_base = 0;
    return MG_OK;
}

// structure test for the SIMPLE kernel specialisation (see substep<>)
bool config_is_simple(const mg_quadrotor_config *c) {
    for (int r = 0; r < 3; ++r)
        for (int col = 0; col < 3; ++col)
            if (r != col && (c->drag_f[3 * r + col] != 0.0f || c->drag_m[3 * r + col] != 0.0f ||
                             c->inertia[3 * r + col] != 0.0f))
                return false;
    for (int i = 0; i < 3; ++i)
        if (c->gravity_center[i] != 0.0f) return false;
    for (int i = 0; i < 4; ++i)
        if (c->prop_coord[3 * i + 2] != 0.0f) return false;
    return c->ct2 == 0.0;
}

int check_state(const mg_quadrotor_state *s) {
    if (!s->pos || !s->vel || !s->omega || !s->propw || !s->rot || !s->ct)
        return mg::set_error(MG_ERR_NULL_POINTER, "mg_quadrotor_state has a NULL array");
    return MG_OK;
}

// What mg_quadrotor_plan holds (caller-owned host memory, see the header): everything a step launch needs
// except the per-call I/O pointers.
struct Plan {
    uint32_t magic;
    int32_t n, device, simple;
    QuadK k;
    mg_quadrotor_state st;
};
constexpr uint32_t PLAN_MAGIC = 0x4d475150u;   // "MGQP"
static_assert(sizeof(Plan) <= sizeof(mg_quadrotor_plan), "mg_quadrotor_plan is too small for the folded constants");

int make_plan(Plan *p, const mg_quadrotor_config *cfg, const mg_quadrotor_autoreset *ar, int32_t n,
              const mg_quadrotor_state *state) {
    MG_REQUIRE_PTR(cfg);
    MG_REQUIRE_PTR(state);
    if (n <= 0) return mg::set_error(MG_ERR_BAD_SIZE, "n_envs=%d", n);
    if (int rc = check_state(state)) return rc;
    if (int rc = fold_config(cfg, &p->k)) return rc;
    if (ar != nullptr) {
        if (state->episode == nullptr)
            return mg::set_error(MG_ERR_NULL_POINTER, "fused auto-reset needs mg_quadrotor_state.episode");
        QuadK &k = p->k;
        k.auto_reset = 1;
        for (int i = 0; i < 3; ++i) { k.init_v_base[i] = ar->init_velocity[i]; k.init_w_base[i] = ar->init_angular_velocity[i]; }
        k.init_v_noisy = ar->init_velocity_noisy;
        k.init_w_noisy = ar->init_angular_velocity_noisy;
        k.seed = ar->seed;
        k.env_id_base = ar->env_id_base;
    }
    p->magic = PLAN_MAGIC;
    p->n = n;
    p->device = mg::device_of(state->pos);
    p->simple = (config_is_simple(cfg) && cfg->task != MG_QUADROTOR_TASK_VELOCITY_CONTROL) ? 1 : 0;
    p->st = *state;
    return MG_OK;
}

int launch_plan(const Plan *p, int32_t n_steps, const float *action, float *obs, float *reward, double *reward64,
                uint8_t *done, uint8_t *failed, void *stream) {
    MG_REQUIRE_PTR(action);
    MG_REQUIRE_PTR(obs);
    MG_REQUIRE_PTR(done);
    if (n_steps <= 0) return mg::set_error(MG_ERR_BAD_SIZE, "n_steps=%d", n_steps);
    mg::DeviceGuard guard(p->device);
    StepIO io{action, obs, reward, reward64, done, failed};
    const int n = p->n, grid = (n + BLOCK - 1) / BLOCK;
    if (p->simple)
        hipLaunchKernelGGL(quadrotor_step_kernel<true>, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, p->k, p->st,
                           io, n, n_steps);
    else
        hipLaunchKernelGGL(quadrotor_step_kernel<false>, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, p->k, p->st,
                           io, n, n_steps);
    return mg::check_launch("quadrotor_step_kernel");
}

int launch_steps(const mg_quadrotor_config *cfg, int32_t n, int32_t n_steps, const mg_quadrotor_state *state,
                 const float *action, float *obs, float *reward, double *reward64, uint8_t *done,
                 uint8_t *failed, void *stream, const mg_quadrotor_autoreset *ar = nullptr) {
    MG_REQUIRE_PTR(cfg);
    MG_REQUIRE_PTR(state);
    MG_REQUIRE_PTR(action);
    MG_REQUIRE_PTR(obs);
    MG_REQUIRE_PTR(done);
    Plan p;
    if (int rc = make_plan(&p, cfg, ar, n, state)) return rc;
    return launch_plan(&p, n_steps, action, obs, reward, reward64, done, failed, stream);
}

}  // namespace

extern "C" int mg_quadrotor_default_config(mg_quadrotor_config *c) {
    MG_REQUIRE_PTR(c);
    // metagym/quadrotor/config.json:1-59 and Quadrotor.__init__ defaults env.py:46-114
    *c = mg_quadrotor_config{};
    c->precision = 0.001;
    c->quality = 0.5;
    c->ct0 = 1.538e-5; c->ct1 = -2.5e-4; c->ct2 = 0.0;
    c->mm = 0.010; c->jm = 2.573e-4; c->ra = 0.2010; c->phi = 0.017242179827506;
    c->fail_velocity = 100.0; c->fail_w = 1000.0; c->fail_range = 1000.0;
    c->min_voltage = 0.10; c->max_voltage = 15.0;
    c->dt = 0.01; c->healthy_reward = 1.0; c->z_offset = 5.0;
    c->x_offset = 50; c->y_offset = 50;
    c->nt = 1000;
    c->task = MG_QUADROTOR_TASK_NO_COLLISION;
    c->inertia[0] = 0.0135f; c->inertia[4] = 0.0135f; c->inertia[8] = 0.024f;
    c->drag_m[0] = 0.074f; c->drag_m[4] = 0.074f; c->drag_m[8] = 0.0506f;
    c->drag_f[0] = 0.12f; c->drag_f[4] = 0.12f; c->drag_f[8] = 0.10f;
    const float pc[12] = {0.18f, 0.18f, 0.f, -0.18f, 0.18f, 0.f, -0.18f, -0.18f, 0.f, 0.18f, -0.18f, 0.f};
    for (int i = 0; i < 12; ++i) c->prop_coord[i] = pc[i];
    c->map_d = nullptr; c->map_h = 100; c->map_w = 100;
    return MG_OK;
}

extern "C" int mg_quadrotor_velocity_targets(const mg_quadrotor_config *cfg, int32_t nt, const float *actions_d,
                                             float *targets_d, void *stream) {
    MG_REQUIRE_PTR(cfg);
    MG_REQUIRE_PTR(actions_d);
    MG_REQUIRE_PTR(targets_d);
    if (nt <= 0) return mg::set_error(MG_ERR_BAD_SIZE, "nt=%d", nt);
    QuadK k;
    if (int rc = fold_config(cfg, &k, false)) return rc;
    mg::DeviceGuard guard(mg::device_of(targets_d));
    hipLaunchKernelGGL(quadrotor_targets_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, k, nt, actions_d, targets_d);
    return mg::check_launch("quadrotor_targets_kernel");
}

extern "C" int mg_quadrotor_reset(const mg_quadrotor_config *cfg, int32_t n, const mg_quadrotor_state *state,
                                  const uint8_t *mask, const double *init_vel, const double *init_omega,
                                  float *obs, void *stream) {
    MG_REQUIRE_PTR(cfg);
    MG_REQUIRE_PTR(state);
    if (n <= 0) return mg::set_error(MG_ERR_BAD_SIZE, "n_envs=%d", n);
    if (int rc = check_state(state)) return rc;
    QuadK k;
    if (int rc = fold_config(cfg, &k)) return rc;
    const int grid = (n + BLOCK - 1) / BLOCK;
    mg::DeviceGuard guard(mg::device_of(state->pos));
    hipLaunchKernelGGL(quadrotor_reset_kernel, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, k, *state, mask,
                       init_vel, init_omega, obs, n);
    return mg::check_launch("quadrotor_reset_kernel");
}

extern "C" int mg_quadrotor_step(const mg_quadrotor_config *cfg, int32_t n, const mg_quadrotor_state *state,
                                 const float *action, float *obs, float *reward, double *reward64,
                                 uint8_t *done, uint8_t *failed, void *stream) {
    return launch_steps(cfg, n, 1, state, action, obs, reward, reward64, done, failed, stream);
}

extern "C" int mg_quadrotor_step_autoreset(const mg_quadrotor_config *cfg, int32_t n, int32_t n_steps,
                                           const mg_quadrotor_state *state, const mg_quadrotor_autoreset *ar,
                                           const float *action, float *obs, float *reward, double *reward64,
                                           uint8_t *done, uint8_t *failed, void *stream) {
    MG_REQUIRE_PTR(ar);
    return launch_steps(cfg, n, n_steps, state, action, obs, reward, reward64, done, failed, stream, ar);
}

extern "C" int mg_quadrotor_rollout(const mg_quadrotor_config *cfg, int32_t n, int32_t n_steps,
                                    const mg_quadrotor_state *state, const float *action, float *obs,
                                    float *reward, double *reward64, uint8_t *done, uint8_t *failed,
                                    void *stream) {
    return launch_steps(cfg, n, n_steps, state, action, obs, reward, reward64, done, failed, stream);
}

extern "C" int mg_quadrotor_plan_init(mg_quadrotor_plan *plan, const mg_quadrotor_config *cfg,
                                      const mg_quadrotor_autoreset *ar, int32_t n_envs,
                                      const mg_quadrotor_state *state) {
    MG_REQUIRE_PTR(plan);
    Plan *p = reinterpret_cast<Plan *>(plan);
    p->magic = 0;
    return make_plan(p, cfg, ar, n_envs, state);
}

extern "C" int mg_quadrotor_plan_step(const mg_quadrotor_plan *plan, int32_t n_steps, const float *action, float *obs,
                                      float *reward, double *reward64, uint8_t *done, uint8_t *failed, void *stream) {
    MG_REQUIRE_PTR(plan);
    const Plan *p = reinterpret_cast<const Plan *>(plan);
    if (p->magic != PLAN_MAGIC) return mg::set_error(MG_ERR_BAD_CONFIG, "mg_quadrotor_plan is not initialised");
    return launch_plan(p, n_steps, action, obs, reward, reward64, done, failed, stream);
}
