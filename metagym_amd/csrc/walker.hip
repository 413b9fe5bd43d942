// walker.hip — batched MetaLocomotion walker (humanoid / ant) engine for gfx950.
//
// Replaces, for N environments per launch, WalkerBaseEnv.step
// (metagym/metalocomotion/envs/utils/walker_base_env.py:43-82):
//   robot.apply_action        humanoids.py:50-54 / walker_base.py:26-29   (motor torques)
//   scene.global_step         scene_bases.py:45-50 -> pybullet.stepSimulation()   (substep() below)
//   robot.calc_state          walker_base.py:31-64, robot_bases.py:317-332        (observe())
//   alive / progress / limits walker_base_env.py:46-82, walker_base.py:66-82      (finish)
//
// PARITY UNPINNED for the physics: the reference delegates it to PyBullet, which is not in the
// reference tree. substep() is a from-scratch reduced-coordinate multibody step with the
// reference's parameters (4 x 5 ms semi-implicit Euler, 5 PGS iterations, ERP 0.9, g = 9.8,
// friction 0.8 x 0.8): joint-space inertia matrix M(q) from world-frame body Jacobians, bias forces
// from a world-frame Newton-Euler pass, Cholesky solve, then projected Gauss-Seidel over ground
// contacts (normal + 2 friction rows per penetrating collision sphere) and joint-limit rows,
// iterated in velocity space (u += M^-1 J_r^T dlambda). oracle/abd.py is the independent numpy
// restatement it is tested against; both are checked by physical invariants.
//
// Mapping: one lane per env, float64. The per-env work set (M: n^2, constraint Jacobians and their
// M^-1 images: 2 * rows * n doubles, n = 6 + joints = 23 for the humanoid) lives in private memory;
// the backend interleaves private arrays across the lanes of a wave, so every lane touching the
// same element index is a coalesced access. This is a first, correctness-oriented mapping
// (DESIGN.md §3.5 lists what a wave-per-env LDS version would change).
#include "mg_common.h"

namespace {

constexpr int WK_BLOCK = 64;
constexpr int NB = MG_WALKER_MAX_BODIES;
constexpr int NJ = MG_WALKER_MAX_JOINTS;
constexpr int NS = MG_WALKER_MAX_SPHERES;
constexpr int ND = 6 + NJ;          // max generalized velocities
constexpr int MAXC = 16;            // max simultaneous ground contacts per env (first MAXC penetrating spheres)
constexpr int MAXR = 3 * MAXC + NJ; // max constraint rows

struct ModelRef {   // offsets into one task's table row
    const double *body_pos, *body_rot, *body_mass, *body_com, *body_inertia;
    const double *joint_anchor, *joint_axis, *joint_lo, *joint_hi, *joint_arm, *joint_damp, *joint_stiff, *motor;
    const double *sph_pos, *sph_r;
};

__device__ __forceinline__ ModelRef model_ref(const mg_walker_topology &tp, const mg_walker_models &ms, int task) {
    const double *p = ms.table + (size_t)task * ms.model_stride;
    const int nb = tp.n_bodies, nj = tp.n_joints, ns = tp.n_spheres;
    ModelRef r;
    r.body_pos = p; p += 3 * nb;
    r.body_rot = p; p += 9 * nb;
    r.body_mass = p; p += nb;
    r.body_com = p; p += 3 * nb;
    r.body_inertia = p; p += 9 * nb;
    r.joint_anchor = p; p += 3 * nj;
    r.joint_axis = p; p += 3 * nj;
    r.joint_lo = p; p += nj;
    r.joint_hi = p; p += nj;
    r.joint_arm = p; p += nj;
    r.joint_damp = p; p += nj;
    r.joint_stiff = p; p += nj;
    r.motor = p; p += nj;
    r.sph_pos = p; p += 3 * ns;
    r.sph_r = p;
    return r;
}

struct V3 { double x, y, z; };
__device__ __forceinline__ V3 v3(double x, double y, double z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(double s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ V3 ld3(const double *p) { return V3{p[0], p[1], p[2]}; }
__device__ __forceinline__ V3 mulMv(const double *R, V3 v) {   // row-major 3x3 times vector
    return V3{R[0] * v.x + R[1] * v.y + R[2] * v.z, R[3] * v.x + R[4] * v.y + R[5] * v.z,
              R[6] * v.x + R[7] * v.y + R[8] * v.z};
}
__device__ __forceinline__ void mulMM(const double *A, const double *B, double *C) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) C[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
}
// Rodrigues rotation matrix about unit axis k by angle t
__device__ __forceinline__ void rodrigues(V3 k, double t, double *R) {
    const double s = sin(t), c = cos(t), v = 1.0 - c;
    R[0] = c + k.x * k.x * v;       R[1] = k.x * k.y * v - k.z * s; R[2] = k.x * k.z * v + k.y * s;
    R[3] = k.y * k.x * v + k.z * s; R[4] = c + k.y * k.y * v;       R[5] = k.y * k.z * v - k.x * s;
    R[6] = k.z * k.x * v - k.y * s; R[7] = k.z * k.y * v + k.x * s; R[8] = c + k.z * k.z * v;
}

struct Env {   // per-lane simulation state
    V3 pos, vel, omega;
    double rot[9];
    double q[NJ], qd[NJ];
};

struct Kin {   // world-frame kinematics of the current configuration
    double R[NB][9];
    V3 o[NB], c[NB];
    V3 p[NJ], a[NJ];
    unsigned mask[NB];   // joints between the base and body b (bit j)
};

__device__ void kinematics(const mg_walker_topology &tp, const ModelRef &m, const Env &s, Kin &k) {
    const int nb = tp.n_bodies, nj = tp.n_joints;
    int j = 0;
    for (int b = 0; b < nb; ++b) {
        double Rc[9];
        V3 oc;
        unsigned mk = 0;
        const int pb = tp.body_parent[b];
        if (pb < 0) {
            for (int i = 0; i < 9; ++i) Rc[i] = s.rot[i];
            oc = s.pos;
        } else {
            mulMM(k.R[pb], m.body_rot + 9 * b, Rc);
            oc = k.o[pb] + mulMv(k.R[pb], ld3(m.body_pos + 3 * b));
            mk = k.mask[pb];
        }
        for (; j < nj && tp.joint_body[j] == b; ++j) {
            const V3 anchor = ld3(m.joint_anchor + 3 * j), axis = ld3(m.joint_axis + 3 * j);
            k.p[j] = oc + mulMv(Rc, anchor);
            k.a[j] = mulMv(Rc, axis);
            double Rj[9], Rn[9];
            rodrigues(axis, s.q[j], Rj);
            mulMM(Rc, Rj, Rn);
            oc = k.p[j] - mulMv(Rn, anchor);
            for (int i = 0; i < 9; ++i) Rc[i] = Rn[i];
            mk |= 1u << j;
        }
        for (int i = 0; i < 9; ++i) k.R[b][i] = Rc[i];
        k.o[b] = oc;
        k.c[b] = oc + mulMv(Rc, ld3(m.body_com + 3 * b));
        k.mask[b] = mk;
    }
}

// column `d` of the Jacobian of point x on a body with joint mask `mk`: velocity of x per unit u_d
__device__ __forceinline__ V3 jac_lin(const Kin &k, unsigned mk, V3 x, int d) {
    if (d < 3) return V3{d == 0 ? 1.0 : 0.0, d == 1 ? 1.0 : 0.0, d == 2 ? 1.0 : 0.0};
    if (d < 6) {
        const V3 e{d == 3 ? 1.0 : 0.0, d == 4 ? 1.0 : 0.0, d == 5 ? 1.0 : 0.0};
        return cross(e, x - k.o[0]);
    }
    const int j = d - 6;
    if (!((mk >> j) & 1u)) return V3{0, 0, 0};
    return cross(k.a[j], x - k.p[j]);
}
__device__ __forceinline__ V3 jac_ang(const Kin &k, unsigned mk, int d) {
    if (d < 3) return V3{0, 0, 0};
    if (d < 6) return V3{d == 3 ? 1.0 : 0.0, d == 4 ? 1.0 : 0.0, d == 5 ? 1.0 : 0.0};
    const int j = d - 6;
    if (!((mk >> j) & 1u)) return V3{0, 0, 0};
    return k.a[j];
}

// M (lower triangle, row-major n x n) and bias h, Featherstone RBDA ch. 3/6 written with world-frame
// Jacobians: M = sum_b m Jv^T Jv + Jw^T I Jw (+ armature), h = sum_b Jv^T m (a_c - g) + Jw^T (I alpha + w x I w)
// where (alpha, a_c) are the velocity-product accelerations (du/dt = 0).
__device__ void mass_and_bias(const mg_walker_topology &tp, const ModelRef &m, const Env &s, const Kin &k,
                              double gravity, double *M, double *h, int n) {
    const int nb = tp.n_bodies, nj = tp.n_joints;
    for (int i = 0; i < n * n; ++i) M[i] = 0.0;
    for (int i = 0; i < n; ++i) h[i] = 0.0;
    V3 fw[NB], fal[NB], fxr[NB], far_[NB];   // frame after each body's joints: w, alpha, ref point, its accel
    int j = 0;
    for (int b = 0; b < nb; ++b) {
        V3 w, al, xr, ar;
        const int pb = tp.body_parent[b];
        if (pb < 0) { w = s.omega; al = v3(0, 0, 0); xr = k.o[0]; ar = v3(0, 0, 0); }
        else { w = fw[pb]; al = fal[pb]; xr = fxr[pb]; ar = far_[pb]; }
        for (; j < nj && tp.joint_body[j] == b; ++j) {
            const V3 r = k.p[j] - xr;
            ar = ar + cross(al, r) + cross(w, cross(w, r));
            xr = k.p[j];
            const V3 wj = s.qd[j] * k.a[j];
            al = al + cross(w, wj);
            w = w + wj;
        }
        fw[b] = w; fal[b] = al; fxr[b] = xr; far_[b] = ar;
        const V3 r = k.c[b] - xr;
        const V3 a_c = ar + cross(al, r) + cross(w, cross(w, r));
        // world inertia Iw = R I R^T
        double RI[9], Iw[9], Rt[9];
        mulMM(k.R[b], m.body_inertia + 9 * b, RI);
        for (int r0 = 0; r0 < 3; ++r0)
            for (int c0 = 0; c0 < 3; ++c0) Rt[3 * r0 + c0] = k.R[b][3 * c0 + r0];
        mulMM(RI, Rt, Iw);
        const double mass = m.body_mass[b];
        const V3 F = mass * (a_c - v3(0, 0, -gravity));
        const V3 Iw_w = mulMv(Iw, w);
        const V3 Nn = mulMv(Iw, al) + cross(w, Iw_w);
        const unsigned mk = k.mask[b];
        for (int d = 0; d < n; ++d) {
            if (d >= 6 && !((mk >> (d - 6)) & 1u)) continue;
            const V3 jv = jac_lin(k, mk, k.c[b], d), jw = jac_ang(k, mk, d);
            h[d] += dot(jv, F) + dot(jw, Nn);
            const V3 mjv = mass * jv, Ijw = mulMv(Iw, jw);
            for (int e = 0; e <= d; ++e) {
                if (e >= 6 && !((mk >> (e - 6)) & 1u)) continue;
                const V3 ev = jac_lin(k, mk, k.c[b], e), ew = jac_ang(k, mk, e);
                M[d * n + e] += dot(mjv, ev) + dot(Ijw, ew);
            }
        }
    }
    for (int jj = 0; jj < nj; ++jj) M[(6 + jj) * n + 6 + jj] += m.joint_arm[jj];
}

// in-place Cholesky of the lower triangle: M = L L^T
__device__ void cholesky(double *M, int n) {
    for (int c = 0; c < n; ++c) {
        double d = M[c * n + c];
        for (int k = 0; k < c; ++k) d -= M[c * n + k] * M[c * n + k];
        d = sqrt(d);
        M[c * n + c] = d;
        const double inv = 1.0 / d;
        for (int r = c + 1; r < n; ++r) {
            double v = M[r * n + c];
            for (int k = 0; k < c; ++k) v -= M[r * n + k] * M[c * n + k];
            M[r * n + c] = v * inv;
        }
    }
}
__device__ void chol_solve(const double *L, int n, double *x) {   // x <- (L L^T)^-1 x
    for (int r = 0; r < n; ++r) {
        double v = x[r];
        for (int k = 0; k < r; ++k) v -= L[r * n + k] * x[k];
        x[r] = v / L[r * n + r];
    }
    for (int r = n - 1; r >= 0; --r) {
        double v = x[r];
        for (int k = r + 1; k < n; ++k) v -= L[k * n + r] * x[k];
        x[r] = v / L[r * n + r];
    }
}

// One physics sub-step. touch[g] = 1 for collision spheres in contact this sub-step.
__device__ void substep(const mg_walker_topology &tp, const ModelRef &m, const mg_walker_params &prm, Env &s,
                        const double *tau_motor, unsigned long long &touch_mask) {
    const int nj = tp.n_joints, ns = tp.n_spheres, n = 6 + nj;
    const double dt = prm.time_step;
    Kin k;
    kinematics(tp, m, s, k);
    double M[ND * ND], h[ND], u[ND];
    mass_and_bias(tp, m, s, k, prm.gravity, M, h, n);
    cholesky(M, n);
    // free motion: u* = u + dt M^-1 (tau - h); explicit damping / spring torques like the oracle
    double rhs[ND];
    for (int d = 0; d < 6; ++d) rhs[d] = -h[d];
    for (int j = 0; j < nj; ++j)
        rhs[6 + j] = tau_motor[j] - m.joint_damp[j] * s.qd[j] - m.joint_stiff[j] * s.q[j] - h[6 + j];
    chol_solve(M, n, rhs);
    u[0] = s.vel.x; u[1] = s.vel.y; u[2] = s.vel.z;
    u[3] = s.omega.x; u[4] = s.omega.y; u[5] = s.omega.z;
    for (int j = 0; j < nj; ++j) u[6 + j] = s.qd[j];
    for (int d = 0; d < n; ++d) u[d] += dt * rhs[d];

    // ---- constraint rows ---------------------------------------------------------------------
    double J[MAXR][ND], W[MAXR][ND];   // W_r = M^-1 J_r^T
    double bias[MAXR], diag[MAXR], lam[MAXR];
    int kind[MAXR], partner[MAXR];
    int nr = 0, ncontacts = 0;
    touch_mask = 0ull;
    for (int g = 0; g < ns && ncontacts < MAXC; ++g) {
        const int b = tp.sphere_body[g];
        const V3 x = k.o[b] + mulMv(k.R[b], ld3(m.sph_pos + 3 * g));
        const double depth = m.sph_r[g] - x.z;
        if (depth > 0.0) {
            const V3 xc{x.x, x.y, 0.0};
            const unsigned mk = k.mask[b];
            for (int d = 0; d < n; ++d) {
                const V3 jc = jac_lin(k, mk, xc, d);
                J[nr][d] = jc.z;
                J[nr + 1][d] = jc.x;
                J[nr + 2][d] = jc.y;
            }
            bias[nr] = prm.erp * depth / dt; kind[nr] = 0; partner[nr] = -1;
            bias[nr + 1] = 0.0; kind[nr + 1] = 1; partner[nr + 1] = nr;
            bias[nr + 2] = 0.0; kind[nr + 2] = 2; partner[nr + 2] = nr;
            nr += 3;
            ++ncontacts;
            touch_mask |= 1ull << g;
        }
    }
    for (int j = 0; j < nj; ++j) {
        double sgn = 0.0, viol = 0.0;
        if (s.q[j] < m.joint_lo[j]) { sgn = 1.0; viol = m.joint_lo[j] - s.q[j]; }
        else if (s.q[j] > m.joint_hi[j]) { sgn = -1.0; viol = s.q[j] - m.joint_hi[j]; }
        if (sgn != 0.0) {
            for (int d = 0; d < n; ++d) J[nr][d] = 0.0;
            J[nr][6 + j] = sgn;
            bias[nr] = prm.limit_erp * viol / dt; kind[nr] = 0; partner[nr] = -1;
            ++nr;
        }
    }
    for (int r = 0; r < nr; ++r) {
        for (int d = 0; d < n; ++d) W[r][d] = J[r][d];
        chol_solve(M, n, W[r]);
        double dd = 0.0;
        for (int d = 0; d < n; ++d) dd += J[r][d] * W[r][d];
        diag[r] = dd;
        lam[r] = 0.0;
    }
    // projected Gauss-Seidel in velocity space: identical iterates to PGS on A = J M^-1 J^T
    for (int it = 0; it < prm.solver_iterations; ++it)
        for (int r = 0; r < nr; ++r) {
            if (!(diag[r] > 0.0)) continue;
            double jv = 0.0;
            for (int d = 0; d < n; ++d) jv += J[r][d] * u[d];
            double x = lam[r] - (jv - bias[r]) / diag[r];
            if (kind[r] == 0) x = x > 0.0 ? x : 0.0;
            else {
                const double lim = prm.friction * lam[partner[r]];
                x = x < -lim ? -lim : (x > lim ? lim : x);
            }
            const double dl = x - lam[r];
            lam[r] = x;
            for (int d = 0; d < n; ++d) u[d] += W[r][d] * dl;
        }
    // ---- integrate ------------------------------------------------------------------------------
    s.vel = v3(u[0], u[1], u[2]);
    s.omega = v3(u[3], u[4], u[5]);
    for (int j = 0; j < nj; ++j) { s.qd[j] = u[6 + j]; s.q[j] += dt * s.qd[j]; }
    s.pos = s.pos + dt * s.vel;
    const double wn = sqrt(dot(s.omega, s.omega));
    if (wn * dt > 0.0) {
        double Rw[9], Rn[9];
        rodrigues((1.0 / wn) * s.omega, wn * dt, Rw);
        mulMM(Rw, s.rot, Rn);
        for (int i = 0; i < 9; ++i) s.rot[i] = Rn[i];
    }
}

// ---- state I/O ---------------------------------------------------------------------------------------

__device__ void load_env(const mg_walker_state &st, int n_envs, int nj, int e, Env &s) {
    s.pos = v3(st.pos[e], st.pos[n_envs + e], st.pos[2 * (size_t)n_envs + e]);
    s.vel = v3(st.vel[e], st.vel[n_envs + e], st.vel[2 * (size_t)n_envs + e]);
    s.omega = v3(st.omega[e], st.omega[n_envs + e], st.omega[2 * (size_t)n_envs + e]);
    for (int i = 0; i < 9; ++i) s.rot[i] = st.rot[(size_t)i * n_envs + e];
    for (int j = 0; j < nj; ++j) { s.q[j] = st.q[(size_t)j * n_envs + e]; s.qd[j] = st.qd[(size_t)j * n_envs + e]; }
}
__device__ void store_env(const mg_walker_state &st, int n_envs, int nj, int e, const Env &s) {
    st.pos[e] = s.pos.x; st.pos[n_envs + e] = s.pos.y; st.pos[2 * (size_t)n_envs + e] = s.pos.z;
    st.vel[e] = s.vel.x; st.vel[n_envs + e] = s.vel.y; st.vel[2 * (size_t)n_envs + e] = s.vel.z;
    st.omega[e] = s.omega.x; st.omega[n_envs + e] = s.omega.y; st.omega[2 * (size_t)n_envs + e] = s.omega.z;
    for (int i = 0; i < 9; ++i) st.rot[(size_t)i * n_envs + e] = s.rot[i];
    for (int j = 0; j < nj; ++j) { st.q[(size_t)j * n_envs + e] = s.q[j]; st.qd[(size_t)j * n_envs + e] = s.qd[j]; }
}

// WalkerBase.calc_state walker_base.py:31-64. Returns walk_target_dist and joints_at_limit.
__device__ void observe(const mg_walker_topology &tp, const ModelRef &m, const mg_walker_params &prm, const Env &s,
                        const float *feet_contact, float *obs, double &target_dist, int &at_limit) {
    const int nb = tp.n_bodies, nj = tp.n_joints, nf = tp.n_feet;
    Kin k;
    kinematics(tp, m, s, k);
    double sx = 0.0, sy = 0.0;
    for (int b = 0; b < nb; ++b) { sx += k.o[b].x; sy += k.o[b].y; }
    const double cnt = (double)(nb + (prm.floor_in_parts ? 1 : 0));   // the floor link sits at the origin
    const double bx = sx / cnt, by = sy / cnt, z = k.o[0].z;
    const double *R = k.R[0];
    const double roll = atan2(R[7], R[8]);
    double sp = -R[6];
    sp = sp < -1.0 ? -1.0 : (sp > 1.0 ? 1.0 : sp);
    const double pitch = asin(sp);
    const double yaw = atan2(R[3], R[0]);
    const double theta = atan2(prm.walk_target_y - by, prm.walk_target_x - bx);
    const double dx = prm.walk_target_x - bx, dy = prm.walk_target_y - by;
    target_dist = sqrt(dy * dy + dx * dx);
    const double ang = theta - yaw;
    const double c = cos(-yaw), sn = sin(-yaw);
    const double vx = c * s.vel.x - sn * s.vel.y, vy = sn * s.vel.x + c * s.vel.y, vz = s.vel.z;
    auto clip5 = [](float v) { return v < -5.0f ? -5.0f : (v > 5.0f ? 5.0f : v); };
    obs[0] = clip5((float)(z - prm.initial_z));
    obs[1] = clip5((float)sin(ang));
    obs[2] = clip5((float)cos(ang));
    obs[3] = clip5((float)(0.3 * vx));
    obs[4] = clip5((float)(0.3 * vy));
    obs[5] = clip5((float)(0.3 * vz));
    obs[6] = clip5((float)roll);
    obs[7] = clip5((float)pitch);
    at_limit = 0;
    for (int j = 0; j < nj; ++j) {
        const double lo = m.joint_lo[j], hi = m.joint_hi[j];
        const float p = (float)(2 * (s.q[j] - 0.5 * (lo + hi)) / (hi - lo));   // robot_bases.py:317-323
        const float v = (float)(0.1 * s.qd[j]);                               // :327-328
        if (fabsf(p) > 0.99f) ++at_limit;
        obs[8 + 2 * j] = clip5(p);
        obs[9 + 2 * j] = clip5(v);
    }
    for (int f = 0; f < nf; ++f) obs[8 + 2 * nj + f] = clip5(feet_contact[f]);
}

__global__ __launch_bounds__(WK_BLOCK) void walker_step_kernel(mg_walker_topology tp, mg_walker_models ms,
                                                               mg_walker_params prm, mg_walker_state st, int n_envs,
                                                               const float *action, float *obs, float *reward,
                                                               float *rewards5, uint8_t *done) {
    const int e = blockIdx.x * WK_BLOCK + threadIdx.x;
    if (e >= n_envs) return;
    const int nj = tp.n_joints, nf = tp.n_feet, obs_dim = 8 + 2 * nj + nf;
    const ModelRef m = model_ref(tp, ms, st.task_id[e]);
    Env s;
    load_env(st, n_envs, nj, e, s);
    double tau[NJ];
    for (int j = 0; j < nj; ++j) {
        float a = action[(size_t)e * nj + j];
        a = a < -1.0f ? -1.0f : (a > 1.0f ? 1.0f : a);                 // humanoids.py:50-54
        tau[j] = m.motor[j] * (double)a;
    }
    unsigned long long touch = 0ull;
    for (int it = 0; it < prm.frame_skip; ++it) substep(tp, m, prm, s, tau, touch);   // scene_bases.py:45-50
    // calc_state runs before the feet flags are refreshed (walker_base_env.py:46 vs :57-63)
    float fc[MG_WALKER_MAX_FEET];
    for (int f = 0; f < nf; ++f) fc[f] = st.feet_contact[(size_t)f * n_envs + e];
    float ob[8 + 2 * NJ + MG_WALKER_MAX_FEET];
    double dist;
    int at_limit;
    observe(tp, m, prm, s, fc, ob, dist, at_limit);
    for (int f = 0; f < nf; ++f) {
        float c = 0.0f;
        for (int g = 0; g < tp.n_spheres; ++g)
            if (((touch >> g) & 1ull) && tp.sphere_body[g] == tp.foot_body[f]) c = 1.0f;
        st.feet_contact[(size_t)f * n_envs + e] = c;
    }
    const double alive = ((double)ob[0] + prm.initial_z > prm.alive_z) ? prm.alive_bonus : prm.dead_bonus;   // :47
    bool finite = true;
    for (int i = 0; i < obs_dim; ++i) finite = finite && isfinite(ob[i]);
    const double pot_old = st.potential[e];
    const double pot = -dist / (prm.time_step * prm.frame_skip);        // walker_base.py:66-82
    const double progress = pot - pot_old;
    const double limit_cost = prm.joints_at_limit_cost * at_limit;
    st.potential[e] = pot;
    const int steps = st.steps[e] + 1;
    st.steps[e] = steps;
    const bool d = (alive < 0) || !finite || (steps >= prm.max_steps);
    for (int i = 0; i < obs_dim; ++i) obs[(size_t)e * obs_dim + i] = ob[i];
    reward[e] = (float)(alive + progress + 0.0 + limit_cost + 0.0);      // :69-77
    if (rewards5) {
        float *r5 = rewards5 + (size_t)e * 5;
        r5[0] = (float)alive; r5[1] = (float)progress; r5[2] = 0.0f; r5[3] = (float)limit_cost; r5[4] = 0.0f;
    }
    done[e] = (uint8_t)d;
    store_env(st, n_envs, nj, e, s);
}

__global__ __launch_bounds__(WK_BLOCK) void walker_reset_kernel(mg_walker_topology tp, mg_walker_models ms,
                                                                mg_walker_params prm, mg_walker_state st, int n_envs,
                                                                const uint8_t *mask, const double *joint_noise,
                                                                float *obs) {
    const int e = blockIdx.x * WK_BLOCK + threadIdx.x;
    if (e >= n_envs) return;
    if (mask != nullptr && mask[e] == 0) return;
    const int nj = tp.n_joints, nf = tp.n_feet, obs_dim = 8 + 2 * nj + nf;
    const ModelRef m = model_ref(tp, ms, st.task_id[e]);
    Env s;
    s.pos = ld3(m.body_pos);
    for (int i = 0; i < 9; ++i) s.rot[i] = m.body_rot[i];
    s.vel = v3(0, 0, 0);
    s.omega = v3(0, 0, 0);
    for (int j = 0; j < nj; ++j) {
        s.q[j] = joint_noise ? joint_noise[(size_t)j * n_envs + e] : 0.0;   // walker_base.py:15
        s.qd[j] = 0.0;
    }
    float fc[MG_WALKER_MAX_FEET];
    for (int f = 0; f < nf; ++f) { fc[f] = 0.0f; st.feet_contact[(size_t)f * n_envs + e] = 0.0f; }
    float ob[8 + 2 * NJ + MG_WALKER_MAX_FEET];
    double dist;
    int at_limit;
    observe(tp, m, prm, s, fc, ob, dist, at_limit);
    st.potential[e] = -dist / (prm.time_step * prm.frame_skip);
    st.steps[e] = 0;
    if (obs)
        for (int i = 0; i < obs_dim; ++i) obs[(size_t)e * obs_dim + i] = ob[i];
    store_env(st, n_envs, nj, e, s);
}

int check_walker(const mg_walker_topology *tp, const mg_walker_models *ms, const mg_walker_params *prm,
                 const mg_walker_state *st, int n) {
    if (!tp || !ms || !prm || !st) return mg::set_error(MG_ERR_NULL_POINTER, "walker: NULL descriptor");
    if (n <= 0) return mg::set_error(MG_ERR_BAD_SIZE, "n_envs=%d", n);
    if (tp->n_bodies < 1 || tp->n_bodies > NB || tp->n_joints < 0 || tp->n_joints > NJ || tp->n_spheres < 0 ||
        tp->n_spheres > NS || tp->n_feet < 0 || tp->n_feet > MG_WALKER_MAX_FEET)
        return mg::set_error(MG_ERR_BAD_SIZE, "walker topology out of range (bodies %d joints %d spheres %d feet %d)",
                             tp->n_bodies, tp->n_joints, tp->n_spheres, tp->n_feet);
    const int need = 25 * tp->n_bodies + 12 * tp->n_joints + 4 * tp->n_spheres;
    if (!ms->table || ms->n_tasks < 1 || ms->model_stride < need)
        return mg::set_error(MG_ERR_BAD_SIZE, "walker model table: stride %d < %d", ms->model_stride, need);
    if (!st->task_id || !st->pos || !st->rot || !st->vel || !st->omega || !st->q || !st->qd || !st->potential ||
        !st->feet_contact || !st->steps)
        return mg::set_error(MG_ERR_NULL_POINTER, "mg_walker_state has a NULL array");
    if (!(prm->time_step > 0) || prm->frame_skip < 1 || prm->solver_iterations < 0)
        return mg::set_error(MG_ERR_BAD_CONFIG, "walker params");
    return MG_OK;
}

}  // namespace

extern "C" int mg_walker_reset(const mg_walker_topology *tp, const mg_walker_models *ms, const mg_walker_params *prm,
                               int32_t n, const mg_walker_state *st, const uint8_t *mask, const double *joint_noise,
                               float *obs, void *stream) {
    if (int rc = check_walker(tp, ms, prm, st, n)) return rc;
    hipLaunchKernelGGL(walker_reset_kernel, dim3((n + WK_BLOCK - 1) / WK_BLOCK), dim3(WK_BLOCK), 0, (hipStream_t)stream,
                       *tp, *ms, *prm, *st, n, mask, joint_noise, obs);
    return mg::check_launch("walker_reset_kernel");
}

extern "C" int mg_walker_step(const mg_walker_topology *tp, const mg_walker_models *ms, const mg_walker_params *prm,
                              int32_t n, const mg_walker_state *st, const float *action, float *obs, float *reward,
                              float *rewards5, uint8_t *done, void *stream) {
    if (int rc = check_walker(tp, ms, prm, st, n)) return rc;
    MG_REQUIRE_PTR(action);
    MG_REQUIRE_PTR(obs);
    MG_REQUIRE_PTR(reward);
    MG_REQUIRE_PTR(done);
    hipLaunchKernelGGL(walker_step_kernel, dim3((n + WK_BLOCK - 1) / WK_BLOCK), dim3(WK_BLOCK), 0, (hipStream_t)stream,
                       *tp, *ms, *prm, *st, n, action, obs, reward, rewards5, done);
    return mg::check_launch("walker_step_kernel");
}
