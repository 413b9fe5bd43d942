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
#include "mg_philox.h"

// The library is built with -ffp-contract=off because the quadrotor and maze kernels must reproduce NumPy's
// unfused arithmetic bit for bit. This engine has no such constraint (it is compared with its numpy oracle to
// a tolerance, and the reference's physics is PyBullet): let the compiler fuse a*b+c here.
#pragma clang fp contract(fast)

namespace {

constexpr int WK_BLOCK = 64;
constexpr int NB = MG_WALKER_MAX_BODIES;
constexpr int NJ = MG_WALKER_MAX_JOINTS;
constexpr int NS = MG_WALKER_MAX_SPHERES;
constexpr int ND = 6 + NJ;          // max generalized velocities
constexpr int MAXC = 12;            // contacts the solver keeps per env: the MAXC deepest of the candidates
constexpr int LANE_MAXCAND = 48;    // = W_MAXCAND below: contact candidates recorded per sub-step before the selection
constexpr int MAXR = 3 * MAXC + NJ; // max constraint rows

struct ModelRef {   // offsets into one task's table row
    const double *body_pos, *body_rot, *body_mass, *body_com, *body_inertia;
    const double *joint_anchor, *joint_axis, *joint_lo, *joint_hi, *joint_arm, *joint_damp, *joint_stiff, *motor;
    const double *sph_pos, *sph_r;
    const double *geom_p0, *geom_p1, *geom_r;
    const double *sph_margin;     // per-proxy contact margins, behind geom_r in rows that carry them
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
    r.sph_r = p; p += ns;
    r.geom_p0 = p; p += 3 * tp.n_geoms;
    r.geom_p1 = p; p += 3 * tp.n_geoms;
    r.geom_r = p; p += tp.n_geoms;
    r.sph_margin = p;             // (only read when mg_walker_params.sphere_margin_in_table says the rows carry them)
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

// Closest points of two segments (Ericson, Real-Time Collision Detection 5.1.9), spheres included.
__device__ __forceinline__ void segment_closest(V3 p1, V3 q1, V3 p2, V3 q2, V3 &c1, V3 &c2) {
    const V3 d1 = q1 - p1, d2 = q2 - p2, r = p1 - p2;
    const double a = dot(d1, d1), e = dot(d2, d2), f = dot(d2, r), eps = 1e-12;
    double sc, tc;
    if (a <= eps && e <= eps) { c1 = p1; c2 = p2; return; }
    if (a <= eps) { sc = 0.0; tc = fmin(fmax(f / e, 0.0), 1.0); }
    else {
        const double c = dot(d1, r);
        if (e <= eps) { tc = 0.0; sc = fmin(fmax(-c / a, 0.0), 1.0); }
        else {
            const double b = dot(d1, d2), den = a * e - b * b;
            sc = den > eps ? fmin(fmax((b * f - c * e) / den, 0.0), 1.0) : 0.0;
            tc = (b * sc + f) / e;
            if (tc < 0.0) { tc = 0.0; sc = fmin(fmax(-c / a, 0.0), 1.0); }
            else if (tc > 1.0) { tc = 1.0; sc = fmin(fmax((b - c) / a, 0.0), 1.0); }
        }
    }
    c1 = p1 + sc * d1;
    c2 = p2 + tc * d2;
}
__device__ __forceinline__ void tangent_basis(V3 n, V3 &t1, V3 &t2) {
    const V3 ref = fabs(n.x) < 0.9 ? V3{1.0, 0.0, 0.0} : V3{0.0, 1.0, 0.0};
    t1 = cross(n, ref);
    t1 = (1.0 / sqrt(dot(t1, t1))) * t1;
    t2 = cross(n, t1);
}

struct Env {   // per-lane simulation state
    V3 pos, vel, omega;
    double rot[9];
    double q[NJ], qd[NJ];
};

// Joint noise of the fused auto-reset: numpy's uniform(low=-0.1, high=0.1) = low + (high - low) * u
// (walker_base.py:15) with u from Philox4x32-10, counter (global env id, step, joint / 4).
__device__ __forceinline__ double reset_joint_noise(const mg_walker_params &prm, int e, int j) {
    uint32_t r[4];
    const uint64_t gid = prm.env_id_base + (uint64_t)e, step = prm.step_index;
    philox4x32_10((uint32_t)gid, (uint32_t)step, (uint32_t)(step >> 32) ^ ((uint32_t)(gid >> 32) << 8),
                  0x57414c4bu + (uint32_t)(j >> 2), (uint32_t)prm.seed, (uint32_t)(prm.seed >> 32), r);
    return -0.1 + 0.2 * ((double)r[j & 3] * (1.0 / 4294967296.0));
}

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
                              double gravity, double k_lin, double k_ang, double *M, double *h, int n) {
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
        V3 F = mass * (a_c - v3(0, 0, -gravity));
        const V3 Iw_w = mulMv(Iw, w);
        V3 Nn = mulMv(Iw, al) + cross(w, Iw_w);
        const unsigned mk = k.mask[b];
        if (k_lin != 0.0 || k_ang != 0.0) {
            // btMultiBody's velocity damping (mg_walker_params.body_*_damping): force -m v_c (k + k |v_c|) at the centre of mass,
            // torque -I w (k + k |w|); their negatives join the bias wrench. v_c = Jv(c_b) u.
            V3 vc = s.vel + cross(s.omega, k.c[b] - k.o[0]);
            for (int jj = 0; jj < nj; ++jj)
                if ((mk >> jj) & 1u) vc = vc + s.qd[jj] * cross(k.a[jj], k.c[b] - k.p[jj]);
            F = F + (mass * (k_lin + k_lin * sqrt(dot(vc, vc)))) * vc;
            Nn = Nn + (k_ang + k_ang * sqrt(dot(w, w))) * Iw_w;
        }
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
    mass_and_bias(tp, m, s, k, prm.gravity, prm.body_linear_damping, prm.body_angular_damping, M, h, n);
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
    // contact candidates in candidate order (ground per proxy, then self pairs; at most W_MAXCAND), then the MAXC deepest of them
    // in candidate order — the wave kernel's two steps (and oracle/abd.py contact_candidates / select_contacts), serially
    int nr = 0;
    touch_mask = 0ull;
    double qx[LANE_MAXCAND][6], qdepth[LANE_MAXCAND];
    int qid[LANE_MAXCAND][2];
    int ncand = 0;
    for (int g = 0; g < ns; ++g) {
        const int b = tp.sphere_body[g];
        const V3 x = k.o[b] + mulMv(k.R[b], ld3(m.sph_pos + 3 * g));
        const double depth = m.sph_r[g] - x.z;
        if (depth > -(prm.sphere_margin_in_table ? m.sph_margin[g] : prm.contact_margin)) {
            touch_mask |= 1ull << g;                   // every proxy inside the margin: what getContactPoints reports
            if (ncand < LANE_MAXCAND) {
                qx[ncand][0] = x.x; qx[ncand][1] = x.y; qx[ncand][2] = 0.0;
                qid[ncand][0] = g; qid[ncand][1] = -1; qdepth[ncand] = depth;
                ++ncand;
            }
        }
    }
    if (prm.self_collision)
        for (int pr = 0; pr < tp.n_pairs && ncand < LANE_MAXCAND; ++pr) {
            const int ga = tp.pair_a[pr], gb = tp.pair_b[pr], ba = tp.geom_body[ga], bb = tp.geom_body[gb];
            V3 ca, cb;
            segment_closest(k.o[ba] + mulMv(k.R[ba], ld3(m.geom_p0 + 3 * ga)), k.o[ba] + mulMv(k.R[ba], ld3(m.geom_p1 + 3 * ga)),
                            k.o[bb] + mulMv(k.R[bb], ld3(m.geom_p0 + 3 * gb)), k.o[bb] + mulMv(k.R[bb], ld3(m.geom_p1 + 3 * gb)),
                            ca, cb);
            const V3 dv = ca - cb;
            const double dist = sqrt(dot(dv, dv)), depth = m.geom_r[ga] + m.geom_r[gb] - dist;
            if (depth > 0.0 && dist > 1e-9) {
                const V3 nrm = (1.0 / dist) * dv;
                const V3 xc = 0.5 * ((ca - m.geom_r[ga] * nrm) + (cb + m.geom_r[gb] * nrm));
                qx[ncand][0] = xc.x; qx[ncand][1] = xc.y; qx[ncand][2] = xc.z; qx[ncand][3] = nrm.x; qx[ncand][4] = nrm.y; qx[ncand][5] = nrm.z;
                qid[ncand][0] = ba; qid[ncand][1] = bb; qdepth[ncand] = depth;
                ++ncand;
            }
        }
    for (int c = 0; c < ncand; ++c) {
        if (ncand > MAXC) {
            int rank = 0;
            const double kc = floor(qdepth[c] * 1048576.0);      // (depths on a 2^-20 m grid: see the wave kernel's selection)
            for (int o = 0; o < ncand; ++o) { const double ko = floor(qdepth[o] * 1048576.0); rank += (ko > kc || (ko == kc && o < c)) ? 1 : 0; }
            if (rank >= MAXC) continue;
        }
        const V3 xc{qx[c][0], qx[c][1], qx[c][2]};
        const double depth = qdepth[c];
        if (qid[c][1] == -1) {                         // ground: point on the plane under the proxy
            const unsigned mk = k.mask[tp.sphere_body[qid[c][0]]];
            for (int d = 0; d < n; ++d) {
                const V3 jc = jac_lin(k, mk, xc, d);
                J[nr][d] = jc.z;
                J[nr + 1][d] = jc.x;
                J[nr + 2][d] = jc.y;
            }
        } else {
            const V3 nrm{qx[c][3], qx[c][4], qx[c][5]};
            const int ba = qid[c][0], bb = qid[c][1];
            V3 t1, t2;
            tangent_basis(nrm, t1, t2);
            for (int d = 0; d < n; ++d) {
                const V3 jd = jac_lin(k, k.mask[ba], xc, d) - jac_lin(k, k.mask[bb], xc, d);
                J[nr][d] = dot(nrm, jd);
                J[nr + 1][d] = dot(t1, jd);
                J[nr + 2][d] = dot(t2, jd);
            }
        }
        const int fk = qid[c][1] == -1 ? 1 : 3;
        bias[nr] = (depth >= 0.0 ? prm.erp * depth : depth) / dt; kind[nr] = 0; partner[nr] = -1;
        bias[nr + 1] = 0.0; kind[nr + 1] = fk; partner[nr + 1] = nr;
        bias[nr + 2] = 0.0; kind[nr + 2] = fk == 1 ? 2 : 3; partner[nr + 2] = nr;
        nr += 3;
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
                const double lim = (kind[r] == 3 ? prm.self_friction : prm.friction) * lam[partner[r]];
                x = x < -lim ? -lim : (x > lim ? lim : x);
            }
            const double dl = x - lam[r];
            lam[r] = x;
            for (int d = 0; d < n; ++d) u[d] += W[r][d] * dl;
        }
    // ---- integrate ------------------------------------------------------------------------------
    if (prm.max_coordinate_velocity > 0.0)        // btMultiBody::applyDeltaVeeMultiDof's clamp (mg_walker_params.max_coordinate_velocity)
        for (int d = 0; d < n; ++d) u[d] = fmin(fmax(u[d], -prm.max_coordinate_velocity), prm.max_coordinate_velocity);
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

// ---- the reference's Python-side arithmetic (pinned by tests/golden/walker_rules.npz) -------------------
// Entries of `robot.parts` that report body b's frame: the base once, every other body once per hinge joint it
// carries (a body with k joints is k links in Bullet's MJCF import, the intermediates massless) or once if it has
// none (fixed joint). walker_base.py:39-41 averages x and y over ALL parts.
__device__ __forceinline__ int part_weight(const mg_walker_topology &tp, int b) {
    if (b == 0) return 1;
    int c = 0;
    for (int j = 0; j < tp.n_joints; ++j) c += tp.joint_body[j] == b;
    return c > 0 ? c : 1;
}
// Humanoid.apply_action humanoids.py:50-54: `1 * power * 0.41 * np.clip(a[i], -1, +1)` with a float32 action is a
// float32 product under NumPy-2 promotion (python floats are weak); WalkerBase.apply_action walker_base.py:26-29
// (the ant) calls float() on the clipped action first and multiplies in float64.
__device__ __forceinline__ double motor_torque(const mg_walker_params &prm, double gain, float a) {
    a = a < -1.0f ? -1.0f : (a > 1.0f ? 1.0f : a);
    return prm.torque_f32 ? (double)((float)gain * a) : gain * (double)a;
}
// In-launch actuators (mg_walker_params.actuation): the reference's PD motor model, laikago_motor.py:157-168, in the exact
// expression order of mg_a1_apply_action (a1.hip) — contraction off here, so the two files produce the same bits.
#pragma clang fp contract(off)
__device__ __forceinline__ double actuator_torque(const mg_walker_params &prm, int j, double q, double qd, double cmd) {
    if (prm.actuation == 2) return cmd;
    if (prm.actuation == 4) return prm.pd_strength[j] * cmd;                         // TORQUE mode :125-128
    double t = (-1.0 * (prm.pd_kp[j] * (q - cmd)) - prm.pd_kd[j] * (qd - 0.0)) + 0.0;
    t = prm.pd_strength[j] * t;
    return fmin(fmax(t, -1.0 * prm.pd_limit[j]), prm.pd_limit[j]);
}
// The shape-generic kernels' actuator: the lane's own command block read once per launch — POSITION with shared or per-robot
// gains, HYBRID (five numbers per motor), TORQUE, raw. Same expression, same order.
struct ActLane { double q_des, kp, qd_des, kd, extra; };
__device__ __forceinline__ double actuator_torque_lane(const mg_walker_params &prm, int j, double q, double qd, const ActLane &a) {
    if (prm.actuation == 2) return a.q_des;
    if (prm.actuation == 4) return prm.pd_strength[j] * a.q_des;
    double t = (-1.0 * (a.kp * (q - a.q_des)) - a.kd * (qd - a.qd_des)) + a.extra;     // laikago_motor.py:157-158
    t = prm.pd_strength[j] * t;
    return fmin(fmax(t, -1.0 * prm.pd_limit[j]), prm.pd_limit[j]);
}
#pragma clang fp contract(fast)

// walker_base_env.py:47 `alive_bonus(state[0] + initial_z, ...)`: state[0] is float32; the humanoid's initial_z is
// the python float 0.8 (humanoids.py:48) -> float32 sum; the ant's came out of calc_state as a float64 -> float64 sum.
__device__ __forceinline__ double alive_height(const mg_walker_params &prm, float obs0) {
    return prm.height_f32 ? (double)(obs0 + (float)prm.initial_z) : (double)obs0 + prm.initial_z;
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
    int parts = prm.floor_in_parts ? 1 : 0;                           // the floor link sits at the origin
    for (int b = 0; b < nb; ++b) {
        const int w = part_weight(tp, b);
        sx += w * k.o[b].x; sy += w * k.o[b].y;
        parts += w;
    }
    const double cnt = (double)parts;
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
        tau[j] = motor_torque(prm, m.motor[j], action[(size_t)e * nj + j]);
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
    for (int f = 0; f < nf; ++f) {        // which flag a proxy reports to: mg_walker_topology.sphere_foot, like the wave kernels
        float c = 0.0f;
        for (int g = 0; g < tp.n_spheres; ++g)
            if (((touch >> g) & 1ull) && tp.sphere_foot[g] == f) c = 1.0f;
        st.feet_contact[(size_t)f * n_envs + e] = c;
    }
    if (st.bad_contacts != nullptr) {     // a1.py:314-323 GetBadFootContacts: contact points on links that are no foot
        int bad = 0;
        for (int g = 0; g < tp.n_spheres; ++g) bad += (((touch >> g) & 1ull) && tp.sphere_foot[g] < 0) ? 1 : 0;
        st.bad_contacts[e] = bad;
    }
    const double alive = (alive_height(prm, ob[0]) > prm.alive_z) ? prm.alive_bonus : prm.dead_bonus;   // :47
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
    if (prm.auto_reset && d) {
        // vector-env convention: the returned obs is the first observation of the next episode
        s.pos = ld3(m.body_pos);
        for (int i = 0; i < 9; ++i) s.rot[i] = m.body_rot[i];
        s.vel = v3(0, 0, 0);
        s.omega = v3(0, 0, 0);
        for (int j = 0; j < nj; ++j) { s.q[j] = reset_joint_noise(prm, e, j); s.qd[j] = 0.0; }
        for (int f = 0; f < nf; ++f) { fc[f] = 0.0f; st.feet_contact[(size_t)f * n_envs + e] = 0.0f; }
        observe(tp, m, prm, s, fc, ob, dist, at_limit);
        st.potential[e] = -dist / (prm.time_step * prm.frame_skip);
        st.steps[e] = 0;
    }
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
    if (prm.reset_pos != nullptr)       // the caller's start pose (mg_walker_params.reset_pos / reset_rot)
        s.pos = v3(prm.reset_pos[e], prm.reset_pos[(size_t)n_envs + e], prm.reset_pos[2 * (size_t)n_envs + e]);
    if (prm.reset_rot != nullptr)
        for (int i = 0; i < 9; ++i) s.rot[i] = prm.reset_rot[(size_t)i * n_envs + e];
    s.vel = v3(0, 0, 0);
    s.omega = v3(0, 0, 0);
    for (int j = 0; j < nj; ++j) {
        s.q[j] = joint_noise ? joint_noise[(size_t)j * n_envs + e] : 0.0;   // walker_base.py:15
        s.qd[j] = 0.0;
    }
    float fc[MG_WALKER_MAX_FEET];
    for (int f = 0; f < nf; ++f) { fc[f] = 0.0f; st.feet_contact[(size_t)f * n_envs + e] = 0.0f; }
    if (prm.reset_pos != nullptr || prm.reset_rot != nullptr) {            // a robot placed by its caller: no contact points yet
        if (st.bad_contacts != nullptr) st.bad_contacts[e] = 0;
        if (st.foot_force != nullptr)
            for (int f = 0; f < nf; ++f) st.foot_force[(size_t)f * n_envs + e] = 0.0;
    }
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


// ======================================================================================================
// Wave-per-env mapping (default). One 64-lane wavefront owns one env; its whole work set lives in LDS
// (18.3 KB for the humanoid: kinematics, packed M, h, the whitened constraint rows Jh = J L^-T, the scan tables; blocks whose
// lifetimes do not overlap share storage, see carve()), so eight envs are resident per CU (two waves per SIMD, the register
// file's cap too) and nothing spills to scratch in the sub-step loop. Serial, branchy sections are what cost time here
// (DESIGN.md 3.4): every phase is written as a few lane-parallel passes. Lanes are dealt
//   * hops (a body's fixed offset or one joint's rotation) for the kinematics and the velocity-product frames: scans over the
//     kinematic tree, log2(chain length) rounds (wave_kinematics),
//   * (body, component) items for the subtree sums, generalized coordinates and matrix entries for M, rows/columns for the
//     Cholesky factorisation and the triangular solves,
//   * collision spheres / joints for constraint detection (ballot + popcount gives every constraint
//     its row index in sphere / joint order, the same order the oracle uses),
//   * constraint rows for the whitening Jh = J L^-T (each lane forward-substitutes its own right-hand side),
//   * whitened coordinates for the PGS sweep (Jh_r . y by a wave reduction, y += Jh_r dlambda per lane).
// Same physics as the lane-per-env kernel above (which keeps the plain J / W = M^-1 J^T solver); the two
// agree to round-off.
// ======================================================================================================

constexpr int WV = 64;
// the joint-space inertia matrix and its Cholesky factor are only ever touched in the lower triangle: packed
#define TRI(r, c) ((r) * ((r) + 1) / 2 + (c))
#ifndef MG_W_MAXC
#define MG_W_MAXC 12      // (timing experiments only — profiles/r05/walker_third_wave.txt: fewer kept contacts = a smaller Jh block)
#endif
constexpr int W_MAXC = MG_W_MAXC;   // contacts the solver keeps per env: the W_MAXC deepest of the candidates
constexpr int W_MAXCAND = 48;       // contact candidates recorded per sub-step before the selection (abd.MAX_CANDIDATES, WO_MAX_CANDIDATES)

// Robot shape as template constants (all zero = read the topology at run time). Every MetaLocomotion variant of a
// robot shares one shape (humanoid: 13 bodies, 17 hinges, 29 collision spheres, 17 capsules; ant: 13 / 8 / 25 / 13),
// and with the shape known at compile time every LDS address and every model-table offset below is a literal in the
// instruction. Computed at run time they were ~40 LDS pointers and 18 table pointers held in SGPRs for the whole
// sub-step: 413 SGPR spills to VGPR lanes in the humanoid kernel (v_writelane / v_readlane on the hot path).
// DMP = 1: the tuned kernel also carries btMultiBody's body velocity damping (the `preset="bullet"` world of the MetaLocomotion
// envs): the frames keep v_ref (15 instead of 12 doubles per body), everything else is the tuned kernel.
template <int B, int J, int S, int G, int OVL, int DMP = 0>
struct Shape { static constexpr int nb = B, nj = J, ns = S, ng = G, overlay = OVL, damp = DMP; };
using ShapeAny = Shape<0, 0, 0, 0, -1>;
// shape-generic, but with the joint count fixed (= every slot of the NMAX-slot register arrays in use): the `slot < n`
// tests of the unrolled Cholesky / substitution code fold away (a scalar compare + branch each otherwise)
template <int J>
using ShapeDof = Shape<0, J, 0, 0, -1>;

struct ModelW {   // one task's table row (layout: mg_walker_models in metagym_hip.h); offsets fold when the shape is constant
    const double *p;          // the row in the model table (global memory)
    const double *c;          // its body / joint constants: the same row (tuned kernels) or the launch's LDS copy (shape-generic ones)
    int nb, nj, ns, ng;
    __device__ __forceinline__ const double *body_pos() const { return c; }
    __device__ __forceinline__ const double *body_rot() const { return c + 3 * nb; }
    __device__ __forceinline__ const double *body_mass() const { return c + 12 * nb; }
    __device__ __forceinline__ const double *body_com() const { return c + 13 * nb; }
    __device__ __forceinline__ const double *body_inertia() const { return c + 16 * nb; }
    __device__ __forceinline__ const double *joint_anchor() const { return c + 25 * nb; }
    __device__ __forceinline__ const double *joint_axis() const { return c + 25 * nb + 3 * nj; }
    __device__ __forceinline__ const double *joint_lo() const { return c + 25 * nb + 6 * nj; }
    __device__ __forceinline__ const double *joint_hi() const { return c + 25 * nb + 7 * nj; }
    __device__ __forceinline__ const double *joint_arm() const { return c + 25 * nb + 8 * nj; }
    __device__ __forceinline__ const double *joint_damp() const { return c + 25 * nb + 9 * nj; }
    __device__ __forceinline__ const double *joint_stiff() const { return c + 25 * nb + 10 * nj; }
    __device__ __forceinline__ const double *motor() const { return c + 25 * nb + 11 * nj; }
    __device__ __forceinline__ const double *sph_pos() const { return p + 25 * nb + 12 * nj; }
    __device__ __forceinline__ const double *sph_r() const { return p + 25 * nb + 12 * nj + 3 * ns; }
    __device__ __forceinline__ const double *geom_p0() const { return p + 25 * nb + 12 * nj + 4 * ns; }
    __device__ __forceinline__ const double *geom_p1() const { return p + 25 * nb + 12 * nj + 4 * ns + 3 * ng; }
    __device__ __forceinline__ const double *geom_r() const { return p + 25 * nb + 12 * nj + 4 * ns + 6 * ng; }
    // per-proxy contact margins behind the capsules (rows that carry them: mg_walker_params.sphere_margin_in_table)
    __device__ __forceinline__ const double *sph_margin() const { return p + 25 * nb + 12 * nj + 4 * ns + 7 * ng; }
};

// Sum over the 64 lanes without LDS: four DPP steps fold each 16-lane row (quad_perm xor-1, xor-2,
// row_half_mirror, row_mirror), then the four row totals are fetched with v_readlane.
template <int CTRL>
__device__ __forceinline__ double dpp_add(double v) {
    // (mov_dpp: no `old` operand to initialise — every lane of these permutations has a valid source)
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
    return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_value(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                            __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double wave_sum(double v) {
    v = dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);   // row_half_mirror
    v = dpp_add<0x140>(v);   // row_mirror
    return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}
// wave-wide minimum / maximum, the result in every lane (four row-local DPP steps, then the four row results)
template <int CTRL, bool MAX>
__device__ __forceinline__ double dpp_minmax(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
    const double o = __hiloint2double(hi, lo);
    return MAX ? fmax(v, o) : fmin(v, o);
}
template <bool MAX>
__device__ __forceinline__ double wave_minmax(double v) {
    v = dpp_minmax<0xB1, MAX>(v);
    v = dpp_minmax<0x4E, MAX>(v);
    v = dpp_minmax<0x141, MAX>(v);
    v = dpp_minmax<0x140, MAX>(v);
    const double a = lane_value(v, 0), b = lane_value(v, 16), c = lane_value(v, 32), d = lane_value(v, 48);
    return MAX ? fmax(fmax(a, b), fmax(c, d)) : fmin(fmin(a, b), fmin(c, d));
}
// same when only lanes 0..31 can hold non-zero terms (at most 32 generalized coordinates)
__device__ __forceinline__ double wave_sum32(double v) {
    v = dpp_add<0xB1>(v);
    v = dpp_add<0x4E>(v);
    v = dpp_add<0x141>(v);
    v = dpp_add<0x140>(v);
    return lane_value(v, 0) + lane_value(v, 16);
}

// One wavefront = one workgroup: LDS operations of a wave retire in program order, so ordering between
// lanes needs no s_barrier and no counter drain — only a compiler barrier so accesses are not reordered.
#define WSYNC()                                                 \
    do {                                                        \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  \
        __builtin_amdgcn_wave_barrier();                        \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  \
    } while (0)

// Optional phase profile (-DMG_WALKER_PROFILE, timing builds only: scripts/walker_phases.py): shader-clock deltas
// between the phase boundaries of the wave kernel, summed over all waves of a launch.
#ifdef MG_WALKER_PROFILE
__device__ unsigned long long mg_walker_phase_cycles[16];
#define PHASE_BEGIN() unsigned long long ph_t0 = __builtin_readcyclecounter()
#define PHASE(i)                                                                          \
    do {                                                                                  \
        const unsigned long long ph_t1 = __builtin_readcyclecounter();                    \
        if (lane == 0) atomicAdd(&mg_walker_phase_cycles[i], ph_t1 - ph_t0);               \
        ph_t0 = ph_t1;                                                                    \
    } while (0)
#elif defined(MG_WALKER_PHASE_MARKS)
// Analysis builds only (scripts/isa_phase_regs.py): a comment line in the assembly at every phase boundary, nothing else
#define PHASE_BEGIN() asm volatile("; MGPHASE begin")
#define PHASE(i) asm volatile("; MGPHASE %0" ::"n"(i))
#else
#define PHASE_BEGIN() do { } while (0)
#define PHASE(i) do { } while (0)
#endif

struct WaveLds {   // pointers into the env's LDS slab
    double *R, *o, *c, *p, *a;               // kinematics
    double *fw, *fal, *fxr, *far_;           // frames after each body's joints
    double *fvr;                             // (shape-generic kernels) velocity of the frame's reference point, for body damping
    double *M, *h, *idg;                     // joint-space inertia (then its packed Cholesky factor), bias, 1/diag(L)
                                             // vector, reciprocal Cholesky diagonal
    double *q, *qd, *tau;
    double *base;                            // pos[3] rot[9] vel[3] omega[3]
    double *J, *bias, *diag, *lam;            // J: constraint rows, whitened in place (Jh = J L^-T)
    double *cx;                              // per contact: ground (x, y, depth) or self (point xc, normal)
    int *mask, *depth, *jstart, *jcount, *kids, *kind, *partner, *csphere, *misc, *dbody;
    int *parent, *sbody, *sfoot;             // topology tables copied out of the kernarg segment: body_parent, sphere_body, sphere_foot
    // scan tables of the kinematics pass (wave_kinematics): a "hop" is one rigid transform of the chain — the fixed
    // offset of body b (hop b) or the rotation of joint j (hop nb + j)
    int *hanc;                               // [rh][nb + nj]: the 2^r-th ancestor hop, -1 past the root
    int *janc;                               // [jr][nj]: the 2^r-th ancestor JOINT of a joint (r = 0: the previous joint on the chain)
    int *bjoint;                             // [nb]: last joint on the chain from the base to body b inclusive, -1 for none
    int *hbody;                              // [nb + nj]: body whose final frame this hop is, -1 for none
    int rh, jr;                              // rounds of the two scans (host: scan_rounds())
};

// LDS layout. The solver works in Cholesky-whitened velocities y = L^T u (M = L L^T): with Jh = J L^-T
//     J M^-1 J^T = Jh Jh^T,   J u = Jh y,   u += M^-1 J^T dl  <=>  y += Jh^T dl,
// so ONE constraint matrix Jh (maxr rows of n doubles) replaces both J and W = M^-1 J^T, and a row costs one
// forward substitution instead of a forward and a backward one. Blocks with disjoint lifetimes share storage:
//  * the Jh block is scratch until the first constraint row is written: the hop transforms and scan buffers of the
//    kinematics pass (12 doubles per hop), then the composite-rigid-body tables (16 doubles per body, 12 per generalized
//    coordinate), then the capsules' world end points of the self-collision pass, all from its start; the velocity-
//    product frames (fw/fal/fxr/far: 12 doubles per body, + v_ref in the shape-generic kernels) at its end
//    (`overlay` says whether the frames fit too; mg_walker_step computes it from the topology);
//  * the solver's reciprocal diagonals and multipliers (diag, lam: 2 maxr doubles) are first written after the
//    constraint rows are complete, when the body frames R / o / c (15 doubles per body) are dead until the
//    next kinematics pass;
//  * M and its Cholesky factor are only ever touched in the lower triangle: packed.
// Humanoid: 33.2 KB in the first layout (4 envs per CU) -> 18.3 KB with the scan tables; the register file (2 waves per
// SIMD) then caps the kernel at 8 envs per CU.
__host__ __device__ inline bool wave_lds_alias2(int nb, int maxr) { return 2 * (size_t)maxr <= 15 * (size_t)nb; }
// fd: doubles per body of velocity-product frames — 12 (w, alpha, x_ref, a_ref), 15 in the shape-generic kernels (+ v_ref)
__host__ __device__ inline size_t wave_lds_doubles(int nb, int nj, int maxr, bool overlay, int fd = 12) {
    const int n = 6 + nj;
    return (size_t)nb * 15 + (size_t)nj * 6 + (overlay ? 0 : fd * (size_t)nb) + (size_t)n * (n + 1) / 2 + 2 * (size_t)n +
           3 * (size_t)nj + 18 + (size_t)maxr * n + (wave_lds_alias2(nb, maxr) ? 1 : 3) * (size_t)maxr +
           6 * (size_t)W_MAXC;
}
__host__ __device__ inline size_t wave_lds_ints(int nb, int nj, int ns, int maxr, int rh, int jr) {
    return 6 * (size_t)nb + 2 * (size_t)ns + 2 * (size_t)maxr + 2 * W_MAXC + 8 + ND +
           (size_t)(rh + 1) * (nb + nj) + (size_t)jr * nj + nb;      // scan tables: hanc, hbody, janc, bjoint
}
// doubles of a model row that precede the collision proxies: body frames, masses, inertias, joint anchors / axes / limits / ...
__host__ __device__ inline size_t wave_model_doubles(int nb, int nj) { return 25 * (size_t)nb + 12 * (size_t)nj; }
// doubles of the Jh block the kinematics pass uses as scratch: one 3x4 transform per hop (then the velocity scans)
__host__ __device__ inline size_t wave_scan_doubles(int nb, int nj) { return 12 * (size_t)(nb + nj); }
// doubles of the Jh block the M / h assembly uses as scratch (composite tables; + the frames when overlaid)
__host__ __device__ inline size_t wave_assembly_doubles(int nb, int nj) { return 16 * (size_t)nb + 12 * (size_t)(6 + nj); }

__device__ __forceinline__ WaveLds carve(unsigned char *smem, int nb, int nj, int ns, int maxr, bool overlay, int fd, int rh, int jr) {
    const int n = 6 + nj;
    WaveLds L;
    double *d = reinterpret_cast<double *>(smem);
    L.R = d; d += 9 * nb; L.o = d; d += 3 * nb; L.c = d; d += 3 * nb; L.p = d; d += 3 * nj; L.a = d; d += 3 * nj;
    double *ne = d;                       // Newton-Euler temporaries: own block, or the tail of the Jh block
    if (!overlay) d += fd * nb;
    L.M = d; d += n * (n + 1) / 2; L.h = d; d += n; L.idg = d; d += n;
    L.q = d; d += nj; L.qd = d; d += nj; L.tau = d; d += nj;
    L.base = d; d += 18;
    L.J = d; d += (size_t)maxr * n;       // constraint rows, whitened in place (Jh)
    if (overlay) ne = d - fd * nb;
    L.fw = ne; ne += 3 * nb; L.fal = ne; ne += 3 * nb; L.fxr = ne; ne += 3 * nb; L.far_ = ne; ne += 3 * nb;
    L.fvr = ne;                           // only touched when fd == 15
    L.bias = d; d += maxr;
    if (wave_lds_alias2(nb, maxr)) { L.diag = L.R; L.lam = L.R + maxr; }
    else { L.diag = d; d += maxr; L.lam = d; d += maxr; }
    L.cx = d; d += 6 * W_MAXC;
    int *i = reinterpret_cast<int *>(d);
    L.mask = i; i += nb; L.depth = i; i += nb; L.jstart = i; i += nb; L.jcount = i; i += nb; L.kids = i; i += nb;
    L.kind = i; i += maxr; L.partner = i; i += maxr; L.csphere = i; i += 2 * W_MAXC; L.misc = i; i += 8; L.dbody = i; i += ND;
    L.parent = i; i += nb; L.sbody = i; i += ns; L.sfoot = i; i += ns;
    L.hanc = i; i += rh * (nb + nj); L.hbody = i; i += nb + nj; L.janc = i; i += jr * nj; L.bjoint = i; i += nb;
    L.rh = rh; L.jr = jr;
    return L;
}

__device__ __forceinline__ V3 ldv(const double *p, int i) { return V3{p[3 * i], p[3 * i + 1], p[3 * i + 2]}; }
__device__ __forceinline__ void stv(double *p, int i, V3 v) { p[3 * i] = v.x; p[3 * i + 1] = v.y; p[3 * i + 2] = v.z; }

// Jacobian columns from LDS-resident kinematics
__device__ __forceinline__ V3 wjac_lin(const WaveLds &L, unsigned mk, V3 x, int d) {
    if (d < 3) return V3{d == 0 ? 1.0 : 0.0, d == 1 ? 1.0 : 0.0, d == 2 ? 1.0 : 0.0};
    if (d < 6) {
        const V3 e{d == 3 ? 1.0 : 0.0, d == 4 ? 1.0 : 0.0, d == 5 ? 1.0 : 0.0};
        return cross(e, x - ldv(L.o, 0));
    }
    const int j = d - 6;
    if (!((mk >> j) & 1u)) return V3{0, 0, 0};
    return cross(ldv(L.a, j), x - ldv(L.p, j));
}
__device__ __forceinline__ V3 wjac_ang(const WaveLds &L, unsigned mk, int d) {
    if (d < 3) return V3{0, 0, 0};
    if (d < 6) return V3{d == 3 ? 1.0 : 0.0, d == 4 ? 1.0 : 0.0, d == 5 ? 1.0 : 0.0};
    const int j = d - 6;
    if (!((mk >> j) & 1u)) return V3{0, 0, 0};
    return ldv(L.a, j);
}

// Rodrigues rotation from the sine and cosine of the angle
__device__ __forceinline__ void rodrigues_sc(V3 k, double s, double c, double *R) {
    const double v = 1.0 - c;
    R[0] = c + k.x * k.x * v;       R[1] = k.x * k.y * v - k.z * s; R[2] = k.x * k.z * v + k.y * s;
    R[3] = k.y * k.x * v + k.z * s; R[4] = c + k.y * k.y * v;       R[5] = k.y * k.z * v - k.x * s;
    R[6] = k.z * k.x * v - k.y * s; R[7] = k.z * k.y * v + k.x * s; R[8] = c + k.z * k.z * v;
}

// `p` again, as slab base + a byte offset the optimiser cannot see into: accesses at constant offsets from the result
// share ONE base register (otherwise every offset beyond an instruction's immediate field gets its own v_add)
__device__ __forceinline__ const double *opaque_base(const double *slab_base, const double *p) {
    unsigned off = (unsigned)((const unsigned char *)p - (const unsigned char *)slab_base);
    asm volatile("" : "+v"(off));
    return (const double *)((const unsigned char *)slab_base + off);
}

// One round of an inclusive scan along the chains of a tree (pointer jumping over static ancestor tables): every lane
// adds the value its 2^r-th ancestor held after the previous round. The values travel through `buf` (K doubles per
// item); reads of a round come before its writes in program order, which is all a single wavefront needs.
template <int K>
__device__ __forceinline__ void scan_add_round(double *buf, int item, int anc, double (&v)[K]) {
    double x[K];
    if (anc >= 0)
        for (int i = 0; i < K; ++i) x[i] = buf[K * anc + i];
    WSYNC();
    if (anc >= 0)
        for (int i = 0; i < K; ++i) { v[i] += x[i]; buf[K * item + i] = v[i]; }
    WSYNC();
}

// Kinematics + velocity-product frames as SCANS over the kinematic tree, lane = hop (the fixed offset of a body or the
// rotation of one joint: nb + nj <= 40 lanes), log2(chain length) rounds instead of one serial pass per tree level:
//   1. every lane builds its hop's transform in the parent frame (joint lanes: sin / cos, Rodrigues, t = anchor - R anchor);
//   2. rh rounds of  T_h <- T_anc(h) o T_h  (3x3 product + rotated offset) leave the WORLD transform after every hop;
//      a joint's anchor and axis are invariant under its own rotation, so p_j = T_j anchor, a_j = R_j axis, and the
//      final hop of a body carries that body's frame (R, o) and centre of mass;
//   3. the velocity-product frames are sums along the chain of per-joint terms — w = w_base + sum qd_k a_k, then
//      alpha = sum w_before x (qd_k a_k) and v_ref = v_base + sum w_before x r_k, then a_ref = sum alpha_before x r_k +
//      w_before x (w_before x r_k), r_k the step from the previous joint anchor — three scans over the joints (jr rounds).
// The level loop this replaced (git history, before round 3's "kinematics ... as scans") issued one full body + joint pass per
// tree level with 1-4 lanes active. forceinline: with two call sites the compiler would otherwise emit a real call, which
// pushes the kernels into scratch.
template <bool VEL>
__device__ __forceinline__ void wave_kinematics(const ModelW &m, const WaveLds &L, int lane, bool with_frames) {
    const int nb = m.nb, nj = m.nj, H = nb + nj;
    double *T = L.J;                          // [H][12] hop transforms (rotation row-major, then offset); scratch of the Jh block
    const bool is_hop = lane < H, is_joint = lane >= nb && lane < H;
    const int j = lane - nb;
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, tr[3] = {0, 0, 0};
    V3 anchor{0, 0, 0}, axis{0, 0, 0};
    PHASE_BEGIN();
    if (is_hop) {
        if (is_joint) {
            anchor = ld3(m.joint_anchor() + 3 * j);
            axis = ld3(m.joint_axis() + 3 * j);
            double sq, cq;
            sincos(L.q[j], &sq, &cq);             // one range reduction for both
            rodrigues_sc(axis, sq, cq, R);
            const V3 ra = mulMv(R, anchor);
            tr[0] = anchor.x - ra.x; tr[1] = anchor.y - ra.y; tr[2] = anchor.z - ra.z;
        } else if (lane == 0) {
            for (int i = 0; i < 9; ++i) R[i] = L.base[3 + i];
            for (int i = 0; i < 3; ++i) tr[i] = L.base[i];
        } else {
            for (int i = 0; i < 9; ++i) R[i] = m.body_rot()[9 * lane + i];
            for (int i = 0; i < 3; ++i) tr[i] = m.body_pos()[3 * lane + i];
        }
        for (int i = 0; i < 9; ++i) T[12 * lane + i] = R[i];
        for (int i = 0; i < 3; ++i) T[12 * lane + 9 + i] = tr[i];
    }
    WSYNC();
    for (int r = 0; r < L.rh; ++r) {
        const int anc = is_hop ? L.hanc[r * H + lane] : -1;
        double A[12];
        if (anc >= 0)
            for (int i = 0; i < 12; ++i) A[i] = T[12 * anc + i];
        WSYNC();
        if (anc >= 0) {
            double Rn[9];
            mulMM(A, R, Rn);
            const V3 tn = mulMv(A, V3{tr[0], tr[1], tr[2]});
            for (int i = 0; i < 9; ++i) R[i] = Rn[i];
            tr[0] = tn.x + A[9]; tr[1] = tn.y + A[10]; tr[2] = tn.z + A[11];
            for (int i = 0; i < 9; ++i) T[12 * lane + i] = R[i];
            for (int i = 0; i < 3; ++i) T[12 * lane + 9 + i] = tr[i];
        }
        WSYNC();
    }
    const V3 o_h{tr[0], tr[1], tr[2]};
    V3 pj{0, 0, 0}, aj{0, 0, 0};
    if (is_joint) {
        pj = o_h + mulMv(R, anchor);
        aj = mulMv(R, axis);
        stv(L.p, j, pj);
        stv(L.a, j, aj);
    }
    const int hb = is_hop ? L.hbody[lane] : -1;
    if (hb >= 0) {
        for (int i = 0; i < 9; ++i) L.R[9 * hb + i] = R[i];
        stv(L.o, hb, o_h);
        stv(L.c, hb, o_h + mulMv(R, ld3(m.body_com() + 3 * hb)));
    }
    WSYNC();
    if (with_frames) {
        double *Sw = T, *Sav = T + 3 * nj, *Sar = T + 9 * nj;       // [nj][3] w sums, [nj][6] alpha | v_ref sums, [nj][3] a_ref sums
        const V3 wb{L.base[15], L.base[16], L.base[17]}, o0 = ldv(L.base, 0);
        const int jp = is_joint ? L.janc[j] : -1;
        const V3 wj = is_joint ? L.qd[j] * aj : V3{0, 0, 0};
        double sw[3] = {wj.x, wj.y, wj.z};
        if (is_joint) stv(Sw, j, wj);
        WSYNC();
        for (int r = 0; r < L.jr; ++r) scan_add_round<3>(Sw, j, is_joint ? L.janc[r * nj + j] : -1, sw);
        V3 wbef = wb, rj{0, 0, 0};
        if (is_joint) {
            if (jp >= 0) wbef = wb + ldv(Sw, jp);
            rj = pj - (jp >= 0 ? ldv(L.p, jp) : o0);
        }
        const V3 dal = cross(wbef, wj), dvr = cross(wbef, rj);
        constexpr int KAV = VEL ? 6 : 3;                            // (v_ref is only kept by the shape-generic kernels)
        double sav[6] = {dal.x, dal.y, dal.z, dvr.x, dvr.y, dvr.z};
        if (is_joint)
            for (int i = 0; i < KAV; ++i) Sav[6 * j + i] = sav[i];
        WSYNC();
        for (int r = 0; r < L.jr; ++r) {
            const int anc = is_joint ? L.janc[r * nj + j] : -1;
            double x[KAV];
            if (anc >= 0)
                for (int i = 0; i < KAV; ++i) x[i] = Sav[6 * anc + i];
            WSYNC();
            if (anc >= 0)
                for (int i = 0; i < KAV; ++i) { sav[i] += x[i]; Sav[6 * j + i] = sav[i]; }
            WSYNC();
        }
        V3 albef{0, 0, 0};
        if (is_joint && jp >= 0) albef = V3{Sav[6 * jp], Sav[6 * jp + 1], Sav[6 * jp + 2]};
        const V3 dar = cross(albef, rj) + cross(wbef, cross(wbef, rj));
        double sar[3] = {dar.x, dar.y, dar.z};
        if (is_joint) stv(Sar, j, dar);
        WSYNC();
        for (int r = 0; r < L.jr; ++r) scan_add_round<3>(Sar, j, is_joint ? L.janc[r * nj + j] : -1, sar);
        if (lane < nb) {
            const int b = lane, k = L.bjoint[b];
            V3 w = wb, al{0, 0, 0}, xr = o0, ar{0, 0, 0}, vr{L.base[12], L.base[13], L.base[14]};
            if (k >= 0) {
                w = wb + ldv(Sw, k);
                al = V3{Sav[6 * k], Sav[6 * k + 1], Sav[6 * k + 2]};
                xr = ldv(L.p, k);
                ar = ldv(Sar, k);
                if (VEL) vr = vr + V3{Sav[6 * k + 3], Sav[6 * k + 4], Sav[6 * k + 5]};
            }
            stv(L.fw, b, w); stv(L.fal, b, al); stv(L.fxr, b, xr); stv(L.far_, b, ar);
            if (VEL) stv(L.fvr, b, vr);
        }
        WSYNC();
    }
    PHASE(11);
}

// Projected Gauss-Seidel in multiplier space, KR rows per sweep fully unrolled (KR >= the row count: a row past the last one has
// idg = bias = lambda = 0 in its lane, so its step is an exact no-op and needs no guard). Delta form, no branch on the row kind:
// every lane forms the change d of ITS multiplier before projection and projects it onto its own bounds; row r's is taken.
// Critical chain per row: fma -> max (-> min) -> readlane -> fmac.
template <int KR, int NRA>
__device__ __forceinline__ void delassus_sweep(const double (&a)[NRA], int iters, int lane, bool fric_l, double idg_l, double c0,
                                               double mu_l, double &g, double &lam) {
    static_assert(KR <= NRA, "rows kept in registers");
    const double INF = __longlong_as_double(0x7ff0000000000000ll);
    double lam_norm = 0.0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < KR; ++r) {
            const double d = fma(-g, idg_l, c0);
            double dl_c;
            if (r % 3 == 0) dl_c = fmax(d, -lam);                // a normal or a joint-limit row: lambda >= 0
            else {                                               // a friction row (its lane < 3 ncont) or a joint-limit row
                const double lim = mu_l * lam_norm;
                const double lo = fric_l ? -lim - lam : -lam, hi = fric_l ? lim - lam : INF;
                dl_c = fmin(fmax(d, lo), hi);
            }
            const double dl = lane_value(dl_c, r);
            g = fma(a[r], dl, g);
            const double xn = lam + dl_c;
            if (r % 3 == 0) lam_norm = lane_value(xn, r);        // (read by the friction rows right after a normal row only)
            lam = lane == r ? xn : lam;
        }
    }
}

template <int NMAX, bool GENERIC, bool VELF>
__device__ __forceinline__ void wave_substep(const mg_walker_topology &tp, const ModelW &m, const mg_walker_params &prm,
                                             const WaveLds &L, int lane, int maxr,
                                             unsigned long long (&touch)[2], const ActLane &act, double *log_row, int n_envs,
                                             double *foot_force, int nf, const double *ext_wrench, const double *terrain,
                                             V3 gvec, double foot_mu) {
    const int nb = m.nb, nj = m.nj, ns = m.ns, n = 6 + nj;
    const double dt = prm.time_step;
    // A fresh copy of the lane id per sub-step: the ~140 lane predicates of the unrolled Cholesky / substitution
    // code (lane == c, k <= lane, ...) are invariant across the sub-step loop, so they were hoisted out of it as
    // 64-bit masks and spilled to VGPR lanes; one v_cmp where it is needed is cheaper than the reload.
    asm volatile("" : "+v"(lane));
    PHASE_BEGIN();
    if (prm.actuation != 0 && lane < nj)      // read by lane 6 + j after the kinematics' barriers
        L.tau[lane] = GENERIC ? actuator_torque_lane(prm, lane, L.q[lane], L.qd[lane], act)
                              : actuator_torque(prm, lane, L.q[lane], L.qd[lane], act.q_des);
    wave_kinematics<VELF>(m, L, lane, true);
    PHASE(0);
    // ---- M and h by the composite-rigid-body algorithm (Featherstone RBDA ch. 6) in world coordinates, all spatial
    //      quantities taken about the base origin O so that subtree sums are plain sums:
    //        body b:  mass m, first moment m r, inertia about O  I_O = I_c + m (|r|^2 1 - r r^T),  r = c_b - O,
    //                 bias wrench (F, N_O = N + r x F) of the velocity-product accelerations;
    //        subtree(b) = own + every descendant, one item per (body, component) over the subtree's bit mask;
    //        dof d on body b_d:  S_d = (v_O, w) = velocity of the point at O and angular velocity per unit u_d,
    //                 F_d = I^c_subtree(b_d) S_d = (m v + w x mr,  mr x v + I_O w),   h_d = S_d . (F, N_O)^c;
    //        M[d][e] = S_e . F_d  when dof e lies on the chain from the base to b_d (e <= d), else 0.
    //      Algebraically the same M = sum_b m Jv^T Jv + Jw^T I_c Jw as the lane kernel / oracle/abd.py assemble from
    //      per-body Jacobians; it replaces 133 (body, dof) Jacobian pairs and a sum over common bodies per matrix
    //      entry (26 % of the sub-step) by 13 + 23 small lane-parallel items and one 6-term dot product per entry.
    double *comp = L.J;                 // [nb][16]: m, m r (3), I_O (xx xy xz yy yz zz), F (3), N_O (3)
    double *Sd = comp + 16 * nb;        // [n][6]
    double *Fd = Sd + 6 * n;            // [n][6]
    const V3 O = ldv(L.o, 0);
    if (lane < nb) {
        const int b = lane;
        const V3 w = ldv(L.fw, b), al = ldv(L.fal, b), xr = ldv(L.fxr, b), ar = ldv(L.far_, b);
        const V3 cb = ldv(L.c, b);
        const V3 rx = cb - xr;
        const V3 a_c = ar + cross(al, rx) + cross(w, cross(w, rx));
        const double *Ib = m.body_inertia() + 9 * b;
        const double *R = L.R + 9 * b;
        double RI[9];
        mulMM(R, Ib, RI);                // I_c = R I_body R^T, symmetric
        const double ixx = RI[0] * R[0] + RI[1] * R[1] + RI[2] * R[2], ixy = RI[0] * R[3] + RI[1] * R[4] + RI[2] * R[5],
                     ixz = RI[0] * R[6] + RI[1] * R[7] + RI[2] * R[8], iyy = RI[3] * R[3] + RI[4] * R[4] + RI[5] * R[5],
                     iyz = RI[3] * R[6] + RI[4] * R[7] + RI[5] * R[8], izz = RI[6] * R[6] + RI[7] * R[7] + RI[8] * R[8];
        auto Ic = [&](V3 x) { return V3{ixx * x.x + ixy * x.y + ixz * x.z, ixy * x.x + iyy * x.y + iyz * x.z,
                                        ixz * x.x + iyz * x.y + izz * x.z}; };
        const double mass = m.body_mass()[b];
        V3 F = mass * (a_c - gvec);              // gvec: the world's gravity acceleration, (0, 0, -gravity) or this robot's own
        V3 N = Ic(al) + cross(w, Ic(w));
        if (VELF && (prm.body_linear_damping != 0.0 || prm.body_angular_damping != 0.0)) {
            // btMultiBody's velocity damping (mg_walker_params.body_*_damping): an external force -m v (k + k |v|) at the
            // centre of mass and torque -I w (k + k |w|), i.e. their negatives join the bias wrench
            const V3 vc = ldv(L.fvr, b) + cross(w, rx);
            const double kl = prm.body_linear_damping, ka = prm.body_angular_damping;
            F = F + (mass * (kl + kl * sqrt(dot(vc, vc)))) * vc;
            N = N + (ka + ka * sqrt(dot(w, w))) * Ic(w);
        }
        const V3 r = cb - O;
        const V3 NO = N + cross(r, F);
        const double rr = dot(r, r);
        double *cp = comp + 16 * b;
        cp[0] = mass; cp[1] = mass * r.x; cp[2] = mass * r.y; cp[3] = mass * r.z;
        cp[4] = ixx + mass * (rr - r.x * r.x); cp[5] = ixy - mass * r.x * r.y; cp[6] = ixz - mass * r.x * r.z;
        cp[7] = iyy + mass * (rr - r.y * r.y); cp[8] = iyz - mass * r.y * r.z; cp[9] = izz + mass * (rr - r.z * r.z);
        cp[10] = F.x; cp[11] = F.y; cp[12] = F.z; cp[13] = NO.x; cp[14] = NO.y; cp[15] = NO.z;
    }
    WSYNC();
    {   // subtree sums, lane = (body, component): every item adds its component over the body's subtree (L.kids: the
        // bodies of the subtree, the body itself included). All items are read before any is written back, so the
        // sums are taken of the bodies' own values and the table is updated in place.
        constexpr int ITEMS = (16 * NB + WV - 1) / WV;
        double acc[ITEMS];
#pragma unroll
        for (int c = 0; c < ITEMS; ++c) {
            const int it = c * WV + lane;
            acc[c] = 0.0;
            if (it < 16 * nb) {
                const double *ci = comp + (it & 15);
                for (unsigned sub = (unsigned)L.kids[it >> 4]; sub != 0; sub &= sub - 1) acc[c] += ci[16 * (__ffs(sub) - 1)];
            }
        }
        WSYNC();
#pragma unroll
        for (int c = 0; c < ITEMS; ++c) {
            const int it = c * WV + lane;
            if (it < 16 * nb) comp[it] = acc[c];
        }
        WSYNC();
    }
    PHASE(1);
    if (lane < n) {
        const int d = lane;
        V3 v{0, 0, 0}, w{0, 0, 0};
        int b = 0;
        if (d < 3) v = V3{d == 0 ? 1.0 : 0.0, d == 1 ? 1.0 : 0.0, d == 2 ? 1.0 : 0.0};
        else if (d < 6) w = V3{d == 3 ? 1.0 : 0.0, d == 4 ? 1.0 : 0.0, d == 5 ? 1.0 : 0.0};
        else {
            w = ldv(L.a, d - 6);
            v = cross(w, O - ldv(L.p, d - 6));
            b = L.dbody[d];
        }
        const double *cp = comp + 16 * b;
        const double mass = cp[0];
        const V3 hc{cp[1], cp[2], cp[3]};
        const V3 pl = mass * v + cross(w, hc);
        const V3 Iw{cp[4] * w.x + cp[5] * w.y + cp[6] * w.z, cp[5] * w.x + cp[7] * w.y + cp[8] * w.z,
                    cp[6] * w.x + cp[8] * w.y + cp[9] * w.z};
        const V3 Lo = cross(hc, v) + Iw;
        double *sp = Sd + 6 * d, *fp = Fd + 6 * d;
        sp[0] = v.x; sp[1] = v.y; sp[2] = v.z; sp[3] = w.x; sp[4] = w.y; sp[5] = w.z;
        fp[0] = pl.x; fp[1] = pl.y; fp[2] = pl.z; fp[3] = Lo.x; fp[4] = Lo.y; fp[5] = Lo.z;
        L.h[d] = dot(v, V3{cp[10], cp[11], cp[12]}) + dot(w, V3{cp[13], cp[14], cp[15]});
    }
    WSYNC();
    for (int t = lane; t < n * (n + 1) / 2; t += WV) {      // packed lower triangle: every lane has an entry
        // row of packed entry t: floor((sqrt(8t + 1) - 1) / 2). With 1.5 under the root a first-of-row t = d(d+1)/2 lands
        // 0.25 / (2d + 1) above 2d + 1 and the entry before it >= 0.06 below: 1-ulp errors of v_sqrt_f32 cannot flip the floor
        const int d = (int)((sqrtf(8.0f * (float)t + 1.5f) - 1.0f) * 0.5f);
        const int e = t - d * (d + 1) / 2;
        double acc = 0.0;
        if (e < 6 || (((unsigned)L.mask[L.dbody[d]] >> (e - 6)) & 1u)) {
            const double *se = Sd + 6 * e, *fd = Fd + 6 * d;
            acc = (se[0] * fd[0] + se[1] * fd[1] + se[2] * fd[2]) + (se[3] * fd[3] + se[4] * fd[4] + se[5] * fd[5]);
        }
        if (d == e && d >= 6) acc += m.joint_arm()[d - 6];
        L.M[t] = acc;                                       // t == TRI(d, e)
    }
    WSYNC();
    PHASE(2);
    // ---- Cholesky in registers, fused with the forward solve of the free motion: lane i keeps row i of the
    //      lower triangle in a fully unrolled NMAX-slot array, so column c needs no LDS at all — L[i][k] is the
    //      lane's own slot k and L[c][k] is lane c's slot k, read with v_readlane (a scalar operand of the FMA).
    //      Slots above the diagonal hold garbage that never reaches a valid entry. The pivot's reciprocal square
    //      root comes from v_rsq_f64 + two Newton steps (the IEEE sqrt followed by an IEEE division was ~550
    //      dependent cycles per column, 16 % of a lone wave's sub-step). As soon as column c is final, L z = b
    //      advances one step with it (z_c = b_c / L_cc; b_i -= L_ic z_c below the diagonal) — no LDS either.
    //      Free motion in whitened coordinates: y* = L^T u + dt L^-1 (tau - h); the backward solve happens once,
    //      after the constraint solver. -----------------------------------------------------------------------
    double u_d = 0.0;   // lane d < n: generalized velocity on entry, whitened velocity y_d from here on
    double x_d = 0.0;
    double idg_d = 0.0; // lane d < n: 1 / L[d][d]
    if (lane < n) {
        const int d = lane;
        x_d = -L.h[d];
        if (d >= 6) {
            const int j = d - 6;
            x_d += L.tau[j] - m.joint_damp()[j] * L.qd[j] - m.joint_stiff()[j] * L.q[j];
            u_d = L.qd[j];
        } else {
            u_d = L.base[d < 3 ? 12 + d : 15 + (d - 3)];
            if (GENERIC && ext_wrench != nullptr) {     // a push on the base body (mg_walker_params.ext_wrench): F on v_O, r x F on omega
                const double *R = L.base + 3;
                const V3 f{ext_wrench[0], ext_wrench[(size_t)n_envs], ext_wrench[2 * (size_t)n_envs]};
                const V3 pl{ext_wrench[3 * (size_t)n_envs], ext_wrench[4 * (size_t)n_envs], ext_wrench[5 * (size_t)n_envs]};
                const V3 F = mulMv(R, f), r = mulMv(R, pl), T = cross(r, F);
                x_d += d == 0 ? F.x : d == 1 ? F.y : d == 2 ? F.z : d == 3 ? T.x : d == 4 ? T.y : T.z;
            }
        }
    }
    const int lrow = lane < n ? lane : n - 1;
    {
        double row[NMAX];
        // unconditional reads: a predicate per slot costs an exec-mask branch each and the 23 masks stay live (spilled)
        // until the write-back below; slots above the diagonal may hold anything
        // (and at unclamped addresses from one per-lane base: slot k > lane reads into the following rows, or just past the
        // factor for the last row of a kernel with spare slots — valid LDS either way)
        const double *Mrow = opaque_base(L.R, L.M + TRI(lrow, 0));
#pragma unroll
        for (int k = 0; k < NMAX; ++k) row[k] = Mrow[k];
        // right-looking: once column c is final every later column takes its rank-1 update at once — the n - c - 1
        // FMAs of a step are independent of each other (the left-looking form chained c dependent FMAs per column)
#pragma unroll
        for (int c = 0; c < NMAX; ++c) {
            if (c < n) {
                const double vc = lane_value(row[c], c), hv = 0.5 * vc;
                double ipiv = __builtin_amdgcn_rsq(vc);
                ipiv = ipiv * fma(-hv * ipiv, ipiv, 1.5);
                ipiv = ipiv * fma(-hv * ipiv, ipiv, 1.5);
                const double lc = lane == c ? vc * ipiv : row[c] * ipiv;       // L[lane][c]
                row[c] = lc;
                constexpr int CG = 6;
                // M[lane][k] -= L[lane][c] L[k][c], six columns at a time: the six broadcasts (v_readlane pairs into SGPRs)
                // first, then the six FMAs — back to back, every FMA would wait two states on its own broadcast (s_nop),
                // and unfenced the scheduler hoists all of a step's v_readlane ahead and spills the SGPRs it ran out of
                if constexpr (GENERIC || NMAX <= 14) {        // (the ant runs three waves per SIMD on 168 VGPRs; shape-generic kernels: interleaved; batching costs them registers and time, A1 2.90 -> 3.00 ms)
#pragma unroll
                    for (int k = c + 1; k < NMAX; ++k) {
                        if (k < n) row[k] -= lc * lane_value(lc, k);
                        if ((k - c) % 6 == 0) __builtin_amdgcn_sched_barrier(0);
                    }
                } else
#pragma unroll
                for (int k0 = c + 1; k0 < NMAX; k0 += CG) {
                    double bc[CG];
#pragma unroll
                    for (int i = 0; i < CG; ++i)
                        if (k0 + i < NMAX && k0 + i < n) bc[i] = lane_value(lc, k0 + i);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < CG; ++i)
                        if (k0 + i < NMAX && k0 + i < n) row[k0 + i] -= lc * bc[i];
                    __builtin_amdgcn_sched_barrier(0);
                }
                idg_d = lane == c ? ipiv : idg_d;     // (no branch inside the column loop: the broadcasts above are
                                                      // convergent and stay put, the FMAs would sink below it)
                const double xr = lane_value(x_d, c) * ipiv;                   // L z = b, one column behind the factor
                x_d = lane == c ? xr : (lane > c ? x_d - lc * xr : x_d);
            }
        }
        if (lane < n) {     // descending: slots above the diagonal alias the diagonal's address and are overwritten by it
            L.idg[lane] = idg_d;
#pragma unroll
            for (int k = NMAX - 1; k >= 0; --k) L.M[TRI(lane, k < lane ? k : lane)] = row[k];
        }
    }
    WSYNC();
    PHASE(3);
    const double *Mcol = opaque_base(L.R, L.M + lrow);      // column `lane` of the packed factor: Mcol[TRI(r, 0)] = L[r][lane]
    {   // y = L^T u: y_d = sum_{r >= d} L[r][d] u_r; the lane's column of L is fetched in one batch of LDS reads
        double col[NMAX];
#pragma unroll
        for (int r = 0; r < NMAX; ++r) {     // (entries above the diagonal are read from wherever TRI(r, lane) lands and dropped)
            const double v = r < n ? Mcol[TRI(r, 0)] : 0.0;
            col[r] = (r < n && r >= lane) ? v : 0.0;
        }
        double y_d = 0.0;
#pragma unroll
        for (int r = 0; r < NMAX; ++r)
            if (r < n) y_d += col[r] * lane_value(u_d, r);
        u_d = lane < n ? y_d + dt * x_d : 0.0;
    }
    PHASE(4);
    // ---- constraint detection ------------------------------------------------------------------------
    //      Two steps (oracle/abd.py contact_candidates / select_contacts). (1) CANDIDATES, lane = collision proxy / geom pair, 64 at
    //      a time, recorded in candidate order (ground per proxy, terrain per proxy, self pairs; ballot + popcount gives the
    //      position) in the idle Jh block behind the capsule end points of the self-collision pass: cx layout (6 doubles), depth,
    //      friction coefficient, and the two ids of csphere; at most cand_cap (W_MAXCAND, later ones dropped). A proxy is a ground /
    //      terrain candidate while depth > -contact_margin (mg_walker_params.contact_margin, Bullet's contact-breaking
    //      threshold), and `touch` = every such proxy, kept by the solver or not — what getContactPoints reports
    //      (walker_base_env.py:57-63). Self pairs: penetration only. (2) SELECTION, lane = candidate: with more than W_MAXC
    //      candidates the W_MAXC deepest stay (rank by depth, ties to the earlier candidate), in candidate order; each kept lane
    //      writes its contact slot and its three rows. Normal-row bias: erp * depth / dt penetrating, the speculative
    //      depth / dt (< 0) inside the margin. With per-proxy friction (mg_walker_params.sphere_friction) a ground friction row
    //      carries its own coefficient in the (otherwise zero) bias slot, kind -1, like a terrain row.
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const int nchunk = GENERIC ? (ns + WV - 1) / WV : 1;
    const bool own_mu = GENERIC && prm.sphere_friction != nullptr;
    const int cand_cap = min(W_MAXCAND, (maxr * n - 8 * m.ng) / 9);       // (mg_walker_step: = W_MAXCAND, or every possible candidate fits)
    double *cand = L.J + 8 * m.ng;                                        // [cand_cap][8]
    int *cand_id = reinterpret_cast<int *>(cand + 8 * cand_cap);          // [cand_cap][2]
    int ncand = 0;
    touch[0] = 0ull;
    touch[1] = 0ull;
    for (int ch = 0; ch < nchunk; ++ch) {
        const int g = ch * WV + lane;
        bool hit = false;
        double sx = 0, sy = 0, depth = 0;
        if (g < ns) {
            const int b = L.sbody[g];
            const V3 xw = ldv(L.o, b) + mulMv(L.R + 9 * b, ld3(m.sph_pos() + 3 * g));
            depth = m.sph_r()[g] - xw.z;
            hit = depth > -(prm.sphere_margin_in_table ? m.sph_margin()[g] : prm.contact_margin);
            sx = xw.x; sy = xw.y;
        }
        const unsigned long long hits = __ballot(hit);
        const int pos = ncand + __popcll(hits & lt_mask);
        if (hit && pos < cand_cap) {
            double *cc = cand + 8 * pos;
            cc[0] = sx; cc[1] = sy; cc[2] = depth; cc[6] = depth;
            // (every friction row carries its mu; a foot proxy takes the robot's own coefficient when there is one: foot_mu >= 0)
            cc[7] = own_mu ? prm.friction * ((foot_mu >= 0.0 && L.sfoot[g] >= 0) ? foot_mu : prm.sphere_friction[g]) : prm.friction;
            cand_id[2 * pos] = g; cand_id[2 * pos + 1] = -1;             // ground contact of proxy g
        }
        touch[GENERIC ? ch : 0] = hits;
        ncand = min(ncand + __popcll(hits), cand_cap);
    }
    // terrain: lane = collision proxy against the static boxes (`terrain`: the batch's one course, or this env's course of
    // the terrain table, mg_walker_params.terrain_id; wave-uniform loop either way: one wave = one env; boxes whose
    // x range misses the chunk's proxies are skipped by the whole wave); per proxy the deepest box, first on ties. Slots
    // after the ground contacts; the friction rows carry the contact's coefficient in their (otherwise zero) bias slot, kind -1.
    if (GENERIC && prm.n_terrain_boxes > 0)
        for (int ch = 0; ch < nchunk; ++ch) {      // (every chunk, even with the candidate buffer full: `touch` counts every proxy inside its margin)
            const int g = ch * WV + lane;
            double rad = 0.0, sx = 0.0, sy = 0.0, sz = 0.0, margin = 0.0;
            if (g < ns) {
                const int b = L.sbody[g];
                const V3 xw = ldv(L.o, b) + mulMv(L.R + 9 * b, ld3(m.sph_pos() + 3 * g));
                rad = m.sph_r()[g];
                margin = prm.sphere_margin_in_table ? m.sph_margin()[g] : prm.contact_margin;
                sx = xw.x; sy = xw.y; sz = xw.z;
            }
            // x range of this chunk's proxies, radius included (y / z ranges as well were measured: they cull nothing more on the
            // reference's courses — rows of boxes along x, a cave the robot does touch — and cost 2 %)
            const bool has = g < ns;
            const double lo = wave_minmax<false>(has ? sx - rad - margin : 1e300), hi = wave_minmax<true>(has ? sx + rad + margin : -1e300);
            double bdepth = -margin, bmu = 0.0;
            V3 bn{0, 0, 1}, bx{0, 0, 0};
            // Broad phase, lane = box (64 per pass): a box whose world x extent misses the x range of this chunk's proxies cannot touch
            // any of them. The survivors (a course is a row of boxes along x: two to five of the reference's 44 ... 68) are then
            // visited in table order by a wave-uniform loop over the ballot's set bits — the same boxes, in the same order, as a
            // serial walk over the table, whose one dependent 128-byte load + test per box was 40 % of the A1's step on `stairstair`.
            for (int b0 = 0; b0 < prm.n_terrain_boxes; b0 += WV) {
              bool near = false;
              if (b0 + lane < prm.n_terrain_boxes) {
                  const double *Bq = terrain + (size_t)MG_WALKER_BOX_DOUBLES * (b0 + lane);
                  const double exq = fabs(Bq[3]) * Bq[12] + fabs(Bq[4]) * Bq[13] + fabs(Bq[5]) * Bq[14];    // world x half extent
                  near = !(Bq[0] + exq < lo || Bq[0] - exq > hi);
              }
              unsigned long long todo = __ballot(near);
              while (todo != 0ull) {
                const int bi = b0 + __builtin_ctzll(todo);
                todo &= todo - 1ull;
                const double *B = terrain + (size_t)MG_WALKER_BOX_DOUBLES * bi;
                const V3 bp{B[0], B[1], B[2]}, bh{B[12], B[13], B[14]};
                if (g < ns) {
                    const V3 rel = V3{sx, sy, sz} - bp;
                    const V3 l{B[3] * rel.x + B[6] * rel.y + B[9] * rel.z, B[4] * rel.x + B[7] * rel.y + B[10] * rel.z,
                               B[5] * rel.x + B[8] * rel.y + B[11] * rel.z};                       // R^T (x - p)
                    V3 c{fmin(fmax(l.x, -bh.x), bh.x), fmin(fmax(l.y, -bh.y), bh.y), fmin(fmax(l.z, -bh.z), bh.z)};
                    const V3 dv = l - c;
                    const double dist2 = dot(dv, dv);
                    double dpt;
                    V3 nl;
                    if (dist2 > 0.0) {
                        const double dist = sqrt(dist2);
                        dpt = rad - dist;
                        nl = (1.0 / dist) * dv;
                    } else {            // centre inside the box: out through the face of least penetration (x, y, z order on ties)
                        const double px = bh.x - fabs(l.x), py = bh.y - fabs(l.y), pz = bh.z - fabs(l.z);
                        const int k = (px <= py && px <= pz) ? 0 : (py <= pz ? 1 : 2);
                        const double lk = k == 0 ? l.x : (k == 1 ? l.y : l.z), sg = lk >= 0.0 ? 1.0 : -1.0;
                        nl = V3{k == 0 ? sg : 0.0, k == 1 ? sg : 0.0, k == 2 ? sg : 0.0};
                        if (k == 0) c.x = sg * bh.x; else if (k == 1) c.y = sg * bh.y; else c.z = sg * bh.z;
                        dpt = rad + (k == 0 ? px : (k == 1 ? py : pz));
                    }
                    if (dpt > bdepth) {
                        bdepth = dpt; bmu = B[15];
                        bn = V3{B[3] * nl.x + B[4] * nl.y + B[5] * nl.z, B[6] * nl.x + B[7] * nl.y + B[8] * nl.z,
                                B[9] * nl.x + B[10] * nl.y + B[11] * nl.z};                        // R n
                        bx = bp + V3{B[3] * c.x + B[4] * c.y + B[5] * c.z, B[6] * c.x + B[7] * c.y + B[8] * c.z,
                                     B[9] * c.x + B[10] * c.y + B[11] * c.z};
                    }
                }
              }
            }
            const bool th = bdepth > -margin;
            const unsigned long long th_mask = __ballot(th);
            const int pos = ncand + __popcll(th_mask & lt_mask);
            if (th && pos < cand_cap) {
                if (own_mu) bmu *= (foot_mu >= 0.0 && L.sfoot[g] >= 0) ? foot_mu : prm.sphere_friction[g];
                double *cc = cand + 8 * pos;
                cc[0] = bx.x; cc[1] = bx.y; cc[2] = bx.z; cc[3] = bn.x; cc[4] = bn.y; cc[5] = bn.z; cc[6] = bdepth; cc[7] = bmu;
                cand_id[2 * pos] = g; cand_id[2 * pos + 1] = -2;                      // proxy g against the world
            }
            touch[ch] |= th_mask;
            ncand = min(ncand + __popcll(th_mask), cand_cap);
        }
    // self-collision: lane = geom pair (in pair order, 64 at a time); slots after the ground and terrain contacts.
    // The capsules' world end points are computed once per geom (lane = geom) into the idle Jh block; a pair then runs
    // the closest-point routine only if its bounding spheres overlap (mid-point distance < half lengths + radii — exactly
    // conservative), and a chunk whose pairs are all apart skips it as a wave (the humanoid's 66 pairs: the second
    // chunk holds two).
    if (prm.self_collision && tp.n_pairs > 0 && ncand < cand_cap) {
        double *seg = L.J;            // [ng][8]: p0 (3), p1 (3), radius, half length
        for (int g = lane; g < m.ng; g += WV) {
            const int b = tp.geom_body[g];
            const V3 ob = ldv(L.o, b);
            const V3 P0 = ob + mulMv(L.R + 9 * b, ld3(m.geom_p0() + 3 * g)), P1 = ob + mulMv(L.R + 9 * b, ld3(m.geom_p1() + 3 * g));
            const V3 d = P1 - P0;
            double *sg = seg + 8 * g;
            sg[0] = P0.x; sg[1] = P0.y; sg[2] = P0.z; sg[3] = P1.x; sg[4] = P1.y; sg[5] = P1.z;
            sg[6] = m.geom_r()[g]; sg[7] = 0.5 * sqrt(dot(d, d));
        }
        WSYNC();
        for (int base = 0; base < tp.n_pairs && ncand < cand_cap; base += WV) {
            const int pr = base + lane;
            bool sh = false, near = false;
            V3 xc{0, 0, 0}, nrm{0, 0, 1};
            double sdepth = 0.0, ra = 0.0, rb = 0.0;
            int ba = 0, bb = 0;
            V3 a0{0, 0, 0}, a1{0, 0, 0}, b0{0, 0, 0}, b1{0, 0, 0};
            if (pr < tp.n_pairs) {
                const int ga = tp.pair_a[pr], gb = tp.pair_b[pr];
                ba = tp.geom_body[ga];
                bb = tp.geom_body[gb];
                const double *sa = seg + 8 * ga, *sb = seg + 8 * gb;
                a0 = ldv(sa, 0); a1 = ldv(sa, 1); b0 = ldv(sb, 0); b1 = ldv(sb, 1);
                ra = sa[6]; rb = sb[6];
                const V3 dm = 0.5 * ((a0 + a1) - (b0 + b1));
                const double reach = (sa[7] + sb[7]) + (ra + rb) + 1e-9;
                near = dot(dm, dm) < reach * reach;
            }
            if (near) {
                V3 ca, cb;
                segment_closest(a0, a1, b0, b1, ca, cb);
                const V3 dv = ca - cb;
                const double dist = sqrt(dot(dv, dv));
                sdepth = ra + rb - dist;
                sh = sdepth > 0.0 && dist > 1e-9;
                if (sh) {
                    nrm = (1.0 / dist) * dv;
                    xc = 0.5 * ((ca - ra * nrm) + (cb + rb * nrm));
                }
            }
            const unsigned long long sh_mask = __ballot(sh);
            const int pos = ncand + __popcll(sh_mask & lt_mask);
            if (sh && pos < cand_cap) {
                double *cc = cand + 8 * pos;
                cc[0] = xc.x; cc[1] = xc.y; cc[2] = xc.z; cc[3] = nrm.x; cc[4] = nrm.y; cc[5] = nrm.z; cc[6] = sdepth; cc[7] = prm.self_friction;
                cand_id[2 * pos] = ba; cand_id[2 * pos + 1] = bb;
            }
            ncand = min(ncand + __popcll(sh_mask), cand_cap);
        }
    }
    // selection: lane = candidate (cand_cap <= 64)
    const int ncont = min(ncand, W_MAXC);
    if (ncand > 0) {
        WSYNC();
        bool keep = lane < ncand;
        const double cdepth = keep ? cand[8 * lane + 6] : 0.0;
        if (ncand > W_MAXC) {           // rank = candidates that go first: deeper ones, and equally deep earlier ones. Depths are
            int rank = 0;               // compared on a 2^-20 m (0.95 um) grid (abd.depth_key): points equally deep by symmetry differ
            const double ckey = floor(cdepth * 1048576.0);        // by round-off between implementations and must tie
            for (int o = 0; o < ncand; ++o) {
                const double okey = lane_value(ckey, o);
                rank += (okey > ckey || (okey == ckey && o < lane)) ? 1 : 0;
            }
            keep = keep && rank < W_MAXC;
        }
        const unsigned long long keep_mask = __ballot(keep);
        if (keep) {
            const int slot = __popcll(keep_mask & lt_mask);
            const double *cc = cand + 8 * lane;
            double *cx = L.cx + 6 * slot;
#pragma unroll
            for (int i = 0; i < 6; ++i) cx[i] = cc[i];
            const int id1 = cand_id[2 * lane + 1];
            L.csphere[2 * slot] = cand_id[2 * lane]; L.csphere[2 * slot + 1] = id1;
            const double mu = cc[7];
            const int fk = id1 >= 0 ? 3 : ((id1 == -2 || own_mu) ? -1 : 1);         // self / own coefficient / the robot's one ground coefficient
            L.bias[3 * slot] = (cdepth >= 0.0 ? prm.erp * cdepth : cdepth) / dt; L.kind[3 * slot] = 0; L.partner[3 * slot] = -1;
            L.bias[3 * slot + 1] = mu; L.kind[3 * slot + 1] = fk; L.partner[3 * slot + 1] = 3 * slot;
            L.bias[3 * slot + 2] = mu; L.kind[3 * slot + 2] = fk == 1 ? 2 : fk; L.partner[3 * slot + 2] = 3 * slot;
        }
    }
    double lsgn = 0.0, viol = 0.0;
    if (lane < nj) {
        const double qj = L.q[lane];
        if (qj < m.joint_lo()[lane]) { lsgn = 1.0; viol = m.joint_lo()[lane] - qj; }
        else if (qj > m.joint_hi()[lane]) { lsgn = -1.0; viol = qj - m.joint_hi()[lane]; }
    }
    const unsigned long long lims = __ballot(lsgn != 0.0);
    const int nr = 3 * ncont + __popcll(lims);
    if (lsgn != 0.0) {
        const int r = 3 * ncont + __popcll(lims & ((1ull << lane) - 1ull));
        // J_r = lsgn * e_(6+joint): kept as (kind 4 / 5 = sign, partner = joint) until the whitening pass builds it
        L.bias[r] = prm.limit_erp * viol / dt; L.kind[r] = lsgn > 0.0 ? 4 : 5; L.partner[r] = lane;
    }
    WSYNC();
    PHASE(5);
    // contact Jacobian rows, lane-strided over (contact, column)
    for (int t = lane; t < ncont * n; t += WV) {
        const int c = t / n, d = t % n;
        const double *cc = L.cx + 6 * c;
        const int other = L.csphere[2 * c + 1];
        if (GENERIC ? other == -1 : other < 0) { // ground: point on the plane under the sphere
            const unsigned mk = (unsigned)L.mask[L.sbody[L.csphere[2 * c]]];
            const V3 jc = wjac_lin(L, mk, V3{cc[0], cc[1], 0.0}, d);
            L.J[(size_t)(3 * c) * n + d] = jc.z;
            L.J[(size_t)(3 * c + 1) * n + d] = jc.x;
            L.J[(size_t)(3 * c + 2) * n + d] = jc.y;
        } else {                                 // self contact: relative velocity of the two bodies at xc; terrain (-2): one body
            const V3 xc{cc[0], cc[1], cc[2]}, nrm{cc[3], cc[4], cc[5]};
            const int b0 = (GENERIC && other == -2) ? L.sbody[L.csphere[2 * c]] : L.csphere[2 * c];
            V3 jd = wjac_lin(L, (unsigned)L.mask[b0], xc, d);
            if (!GENERIC || other >= 0) jd = jd - wjac_lin(L, (unsigned)L.mask[other], xc, d);
            V3 t1, t2;
            tangent_basis(nrm, t1, t2);
            L.J[(size_t)(3 * c) * n + d] = dot(nrm, jd);
            L.J[(size_t)(3 * c + 1) * n + d] = dot(t1, jd);
            L.J[(size_t)(3 * c + 2) * n + d] = dot(t2, jd);
        }
    }
    WSYNC();
    PHASE(6);
    // ---- Jh = J L^-T, lane = row, in place. The right-hand side lives in registers (a fully unrolled
    //      NMAX-slot array): through LDS every step of the substitution would wait on its own previous store.
    //      A joint-limit row starts as +-e_(6+j) and is never materialised before this point -------------------
    static_assert(3 * W_MAXC + NJ <= WV, "one lane per constraint row");
    // (shape-generic instantiations: the whitened row stays in registers past this block — the Delassus sweep below builds its
    // matrix from it)
    constexpr bool ASPACE = true;
#ifndef MG_WALKER_NRA_BIG
#define MG_WALKER_NRA_BIG 42      // the tuned humanoid kernel (see below; experiments: 18 frees 12 VGPRs under a 168-VGPR cap)
#endif
    // Constraint rows the multiplier-space sweep keeps in registers (one row of A per lane, 2 VGPRs per row); more rows take the
    // velocity-space sweep. Round 6: with the contact margin a humanoid lying on the floor has 28 rows per sub-step on average (8.4
    // contacts, profiles/r06/walker_flops.json), over the former 24 — the whole lying batch fell back to the ~25-operation chain per
    // row. Measured on the lying C4_grounded batch / the airborne C4 batch (scripts/bench_walker.py, one MI355X):
    //     24 rows (226 VGPRs) 1.241 / 0.582 ms    30: 1.197 / 0.584    36 (238): 1.111 - 1.123 / 0.586    42 (250): 1.025 - 1.040 / 0.588
    //     48 (256, 2 spills): 1.013 / 0.595
    // 42 rows: -17 % on the contact-rich batch for +1 % on the airborne one, no spill, still two waves per SIMD. The shape-generic
    // 23-slot kernel stays at 24 (it spills from 30 on).
    constexpr int NRA = NMAX <= 18 ? 30 : (GENERIC ? 24 : MG_WALKER_NRA_BIG);
    double w[NMAX];
    if (ASPACE) {
#pragma unroll
        for (int d = 0; d < NMAX; ++d) w[d] = 0.0;
    }
    if (lane < nr) {        // (not a lane-strided loop: its invariant L.M reads would be hoisted into ~500 VGPRs)
        const int r = lane;
        const int rkind = L.kind[r];
        double *jr = L.J + (size_t)r * n;
        if (rkind >= 4) {   // a joint-limit row is written out first (one branch; a select per element cost a branch each)
            const int ldof = 6 + L.partner[r];
            const double lsg = rkind == 4 ? 1.0 : -1.0;
#pragma unroll
            for (int d = 0; d < NMAX; ++d)
                if (d < n) jr[d] = d == ldof ? lsg : 0.0;
        }
#pragma unroll
        for (int d = 0; d < NMAX; ++d) w[d] = d < n ? jr[d] : 0.0;
        // the factor through a base register of its own: its packed offsets then fit the 8-bit fields of ds_read2_b64
        const double *Mp = opaque_base(L.R, L.M);
        double dd = 0.0;
        // (left-looking on purpose: the right-looking form — independent FMAs, one batch of factor reads per column — has fewer
        // instructions and is slower, 0.634 -> 0.664 ms on the humanoid batch)
#pragma unroll
        for (int d = 0; d < NMAX; ++d) {
            if (d < n) {
                double v = w[d];
#pragma unroll
                for (int k = 0; k < d; ++k) v -= Mp[TRI(d, k)] * w[k];
                w[d] = v * L.idg[d];
                dd += w[d] * w[d];
                jr[d] = w[d];
            }
        }
        L.diag[r] = dd > 0.0 ? 1.0 / dd : 0.0;          // reciprocal of Jh_r . Jh_r = J_r M^-1 J_r^T
        L.lam[r] = 0.0;
    }
    WSYNC();
    PHASE(7);
    // ---- projected Gauss-Seidel on the whitened velocity y (wave-uniform row loop) --------------------
    //      The row's Jacobian element and scalars are fetched one row AHEAD, so their LDS latency hides behind
    //      the previous row's reduction; the reduction itself only spans the lanes that hold coordinates.
    //      The multipliers live in registers (lane r holds lambda_r; the current row's is fetched with v_readlane at a scalar
    //      index) and a friction row's partner is always the normal row swept just before it, whose new multiplier is carried
    //      along: no LDS read sits between the reduction and the update any more (two exposed LDS latencies per row).
    //      Shape-generic instantiations with <= NRA rows (the A1: 12 contact rows + its joint limits, 23 iterations x 13
    //      sub-steps) sweep in MULTIPLIER space instead (delassus_sweep): A = Jh Jh^T once (lane j keeps row j in registers, built
    //      from its own whitened row and broadcast LDS reads of the others), g_j = Jh_j . y kept current by g += A[:, r] dlambda_r.
    //      Every lane computes the projected change of ITS multiplier from per-lane data and row r's — whose g is current — is
    //      taken: no reduction, no LDS, no branch on the row kind inside the sweep, and a dependent chain of four operations per row
    //      (the velocity-space row: ~25). A wave runs this engine latency-bound — throughput follows the number of resident
    //      waves, profiles/r04/walker_occupancy.txt — so the chain length is what counts: A1 2.04 -> 1.58 ms per env step.
    //      Same iterates in exact arithmetic (different rounding); y receives sum_k Jh_k^T lambda_k once at the end.
    const bool aspace = ASPACE && nr <= NRA;
    if (ASPACE && nr > 0 && aspace) {
        double a[NRA];
#pragma unroll
        for (int k = 0; k < NRA; ++k) {
            a[k] = 0.0;
            if (k < nr) {
                const double *jk = L.J + (size_t)k * n;
                double acc = 0.0;
#pragma unroll
                for (int d = 0; d < NMAX; ++d)
                    if (d < n) acc += w[d] * jk[d];
                a[k] = acc;
            }
        }
        double g = 0.0;                                              // Jh_lane . y
#pragma unroll
        for (int d = 0; d < NMAX; ++d)
            if (d < n) g += w[d] * lane_value(u_d, d);
        const int nc3 = 3 * ncont;                                   // rows below: (normal, friction, friction) triplets; above: joint limits
        const bool own = lane < nr;
        const double idg_l = own ? L.diag[lane] : 0.0, bias_l = own ? L.bias[lane] : 0.0;
        const bool fric_l = lane < nc3 && lane % 3 != 0;
        const double c0 = fric_l ? 0.0 : bias_l * idg_l;             // d = (bias - g) idg: a friction row's target is zero (its bias slot is mu)
        const double mu_l = fric_l ? bias_l : 0.0;
        double lam = 0.0;
        switch ((nr + 5) / 6) {             // sweep length = the row count rounded up to a multiple of six
        case 1: delassus_sweep<6, NRA>(a, prm.solver_iterations, lane, fric_l, idg_l, c0, mu_l, g, lam); break;
        case 2: delassus_sweep<12, NRA>(a, prm.solver_iterations, lane, fric_l, idg_l, c0, mu_l, g, lam); break;
        case 3: delassus_sweep<18, NRA>(a, prm.solver_iterations, lane, fric_l, idg_l, c0, mu_l, g, lam); break;
        default: delassus_sweep<NRA, NRA>(a, prm.solver_iterations, lane, fric_l, idg_l, c0, mu_l, g, lam); break;
        case 4: delassus_sweep<(NRA < 24 ? NRA : 24), NRA>(a, prm.solver_iterations, lane, fric_l, idg_l, c0, mu_l, g, lam); break;
        }
        for (int k = 0; k < nr; ++k) {                               // y += Jh^T lambda
            const double lk = lane_value(lam, k);
            if (lane < n) u_d += L.J[(size_t)k * n + lane] * lk;
        }
        if (own) L.lam[lane] = lam;                                  // (read by the foot-force block)
    } else
    if (nr > 0) {
        double lamv = 0.0, lam_norm = 0.0;
        // one row: Jh_r . y by a wave reduction, the projected multiplier update, y += Jh_r^T dlambda
        auto row_step = [&](double jh, double idg, double bias, int rkind, int rr) {
            if (!(idg > 0.0)) {
                if (rkind == 0) lam_norm = 0.0;                      // (an empty normal row keeps its zero multiplier)
                return;
            }
            const double jv = NMAX <= 32 ? wave_sum32(jh * u_d) : wave_sum(jh * u_d);   // J_r u = Jh_r . y
            const double lr = lane_value(lamv, rr);
            // a friction row (kinds 1, 2, 3, -1) keeps its coefficient in the bias slot — its velocity target is zero — so the
            // loop needs no per-kind coefficient (it came from the kernarg segment with an s_load + wait per row)
            const bool fric = rkind != 0 && rkind < 4;
            double x = lr - (jv - (fric ? 0.0 : bias)) * idg;
            if (!fric) {
                x = x > 0.0 ? x : 0.0;
                if (rkind == 0) lam_norm = x;
            } else {
                const double lim = bias * lam_norm;
                x = x < -lim ? -lim : (x > lim ? lim : x);
            }
            u_d += jh * (x - lr);                                    // y += Jh_r^T dlambda
            lamv = lane == rr ? x : lamv;
        };
        // the row's Jacobian element and scalars are fetched one row AHEAD into a second register set; the loop is unrolled by
        // two so the sets swap roles instead of being copied (the copies were 8 of the row's ~90 instructions)
        auto fetch = [&](int r, double &jh, double &idg, double &bias, int &kind) {
            jh = lane < n ? L.J[(size_t)r * n + lane] : 0.0;
            idg = L.diag[r]; bias = L.bias[r]; kind = __builtin_amdgcn_readfirstlane(L.kind[r]);
        };
        double jh_a, idg_a, bias_a, jh_b, idg_b, bias_b;
        int kind_a, kind_b;
        fetch(0, jh_a, idg_a, bias_a, kind_a);
        const int sweeps = prm.solver_iterations * nr;
        int r = 0;
        for (int s = 0; s < sweeps; s += 2) {
            const int r1 = r + 1 < nr ? r + 1 : 0;
            fetch(r1, jh_b, idg_b, bias_b, kind_b);
            row_step(jh_a, idg_a, bias_a, kind_a, r);
            if (s + 1 >= sweeps) break;
            const int r2 = r1 + 1 < nr ? r1 + 1 : 0;
            fetch(r2, jh_a, idg_a, bias_a, kind_a);
            row_step(jh_b, idg_b, bias_b, kind_b, r1);
            r = r2;
        }
        if (lane < nr) L.lam[lane] = lamv;                           // (read by the foot-force block)
    }
    PHASE(8);
    if (GENERIC && foot_force != nullptr) {     // normal force per foot (mg_walker_state.foot_force): lane = foot
        WSYNC();                                // lane 0 wrote the multipliers
        if (lane < nf) {
            V3 f{0, 0, 0};
            for (int c = 0; c < ncont; ++c) {
                const int other = L.csphere[2 * c + 1];
                if (other >= 0 || L.sfoot[L.csphere[2 * c]] != lane) continue;      // self contact / another link's proxy
                const double lam_n = L.lam[3 * c];
                f = f + lam_n * (other == -1 ? V3{0, 0, 1} : V3{L.cx[6 * c + 3], L.cx[6 * c + 4], L.cx[6 * c + 5]});
            }
            foot_force[(size_t)lane * n_envs] = sqrt(dot(f, f)) / dt;
        }
    }
    // ---- back to generalized velocities: u = L^-T y (column of L in registers, reciprocal diagonal from the
    //      owning lane: no LDS inside the dependent chain) ------------------------------------------------------
    {
        double col[NMAX];
#pragma unroll
        for (int r = 0; r < NMAX; ++r) {
            const double v = r < n ? Mcol[TRI(r, 0)] : 0.0;
            col[r] = (r < n && r > lane) ? v : 0.0;
        }
#pragma unroll
        for (int r = NMAX - 1; r >= 0; --r) {
            if (r < n) {
                const double xr = lane_value(u_d * idg_d, r);
                u_d = lane == r ? xr : u_d - col[r] * xr;
            }
        }
    }
    // ---- integrate -------------------------------------------------------------------------------------
    if (prm.max_coordinate_velocity > 0.0)        // btMultiBody::applyDeltaVeeMultiDof's clamp (mg_walker_params.max_coordinate_velocity)
        u_d = fmin(fmax(u_d, -prm.max_coordinate_velocity), prm.max_coordinate_velocity);
    if (lane < n) {
        if (lane >= 6) {
            const int j = lane - 6;
            L.qd[j] = u_d;
            L.q[j] += dt * u_d;
        } else if (lane < 3) L.base[12 + lane] = u_d;
        else L.base[15 + (lane - 3)] = u_d;
    }
    WSYNC();
    if (lane == 0) {
        const V3 vel{L.base[12], L.base[13], L.base[14]}, om{L.base[15], L.base[16], L.base[17]};
        L.base[0] += dt * vel.x; L.base[1] += dt * vel.y; L.base[2] += dt * vel.z;
        const double wn = sqrt(dot(om, om));
        if (wn * dt > 0.0) {
            double Rw[9], Rn[9], Ro[9];
            for (int i = 0; i < 9; ++i) Ro[i] = L.base[3 + i];
            double sw, cw;
            sincos(wn * dt, &sw, &cw);
            rodrigues_sc((1.0 / wn) * om, sw, cw, Rw);
            mulMM(Rw, Ro, Rn);
            for (int i = 0; i < 9; ++i) L.base[3 + i] = Rn[i];
        }
    }
    WSYNC();
    PHASE(9);
    if (log_row != nullptr) {     // one Minitaur.GetTrueObservation: q, qd, torque | quaternion (x y z w) | body-frame rate
        if (lane < nj) {
            log_row[(size_t)lane * n_envs] = L.q[lane];
            log_row[(size_t)(nj + lane) * n_envs] = L.qd[lane];
            log_row[(size_t)(2 * nj + lane) * n_envs] = L.tau[lane];
        }
        if (lane == 0) {
            const double *R = L.base + 3;
            const V3 om{L.base[15], L.base[16], L.base[17]};
            double qx, qy, qz, qw;
            const double tr = R[0] + R[4] + R[8];
            if (tr > 0.0) {
                const double s4 = 2.0 * sqrt(tr + 1.0);
                qw = 0.25 * s4; qx = (R[7] - R[5]) / s4; qy = (R[2] - R[6]) / s4; qz = (R[3] - R[1]) / s4;
            } else if (R[0] > R[4] && R[0] > R[8]) {
                const double s4 = 2.0 * sqrt(1.0 + R[0] - R[4] - R[8]);
                qw = (R[7] - R[5]) / s4; qx = 0.25 * s4; qy = (R[1] + R[3]) / s4; qz = (R[2] + R[6]) / s4;
            } else if (R[4] > R[8]) {
                const double s4 = 2.0 * sqrt(1.0 + R[4] - R[0] - R[8]);
                qw = (R[2] - R[6]) / s4; qx = (R[1] + R[3]) / s4; qy = 0.25 * s4; qz = (R[5] + R[7]) / s4;
            } else {
                const double s4 = 2.0 * sqrt(1.0 + R[8] - R[0] - R[4]);
                qw = (R[3] - R[1]) / s4; qx = (R[2] + R[6]) / s4; qy = (R[5] + R[7]) / s4; qz = 0.25 * s4;
            }
            double *r = log_row + (size_t)(3 * nj) * n_envs;
            r[0] = qx; r[(size_t)n_envs] = qy; r[2 * (size_t)n_envs] = qz; r[3 * (size_t)n_envs] = qw;
            // R^T omega
            r[4 * (size_t)n_envs] = R[0] * om.x + R[3] * om.y + R[6] * om.z;
            r[5 * (size_t)n_envs] = R[1] * om.x + R[4] * om.y + R[7] * om.z;
            r[6 * (size_t)n_envs] = R[2] * om.x + R[5] * om.y + R[8] * om.z;
        }
    }
}

// Waves per SIMD: two (<= 256 VGPRs) everywhere but in the tuned ant kernel — its 10.7 KB of LDS let 12 envs reside per CU, so
// it is held to 168 VGPRs for a third wave (its 46 spilled VGPRs all sit after the sub-step loop): 0.402 -> 0.365 ms. The A1's
// 18-slot kernel at 168 VGPRs spills inside the loop and loses (2.31 -> 2.45 ms); the humanoid is LDS-bound at 8 per CU.
#ifndef MG_WALKER_HUM_WAVES
#define MG_WALKER_HUM_WAVES 2     // waves per SIMD the tuned humanoid kernel is compiled for (experiments: 3 = a 168-VGPR cap)
#endif
#ifndef MG_WALKER_A1_WAVES
#define MG_WALKER_A1_WAVES 2      // ... and the 12-hinge quadruped's (<18, ShapeDof<12>>)
#endif
template <int NMAX, class SH>
__global__ __launch_bounds__(WV) __attribute__((amdgpu_waves_per_eu((NMAX <= 14 && SH::nb != 0) ? 3 : (SH::nb != 0 ? MG_WALKER_HUM_WAVES : ((NMAX == 18 && SH::nj == 12) ? MG_WALKER_A1_WAVES : 2))))) void walker_step_wave_kernel(mg_walker_topology tp, mg_walker_models ms,
                                                              mg_walker_params prm, mg_walker_state st, int n_envs,
                                                              int maxr_flags, int scan_rounds, const float *action, float *obs,
                                                              float *reward, float *rewards5, uint8_t *done) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int e = mg::env_of_block(blockIdx.x, n_envs), lane = threadIdx.x;
    const int nb = SH::nb ? SH::nb : tp.n_bodies, nj = SH::nj ? SH::nj : tp.n_joints, ns = SH::nb ? SH::ns : tp.n_spheres;
    const int nf = tp.n_feet, obs_dim = 8 + 2 * nj + nf;
    const double *row = ms.table + (size_t)st.task_id[e] * ms.model_stride;
    // sign of the row-count argument: assembly scratch overlaid on Jh
    const bool overlay = SH::overlay >= 0 ? SH::overlay != 0 : maxr_flags < 0;
    const int maxr = SH::nb ? 3 * W_MAXC + SH::nj : (maxr_flags < 0 ? -maxr_flags : maxr_flags);
    // The slab's base goes through a VGPR the optimiser cannot see into: every LDS access below is then
    // `ds_* v_base offset:<literal>`. Left as the symbol `smem`, each distinct constant address (hundreds in the
    // unrolled Cholesky / whitening loops) was materialised in its own SGPR, hoisted, and spilled to VGPR lanes.
    unsigned slab_off = 0;
    asm volatile("" : "+v"(slab_off));
    unsigned char *slab = smem + slab_off;
    constexpr bool GENERIC = SH::nb == 0;      // shape-generic instantiation: terrain, > 64 proxies, per-proxy friction
    constexpr bool VELF = GENERIC || SH::damp != 0;     // frames keep v_ref: body velocity damping available
    const WaveLds L = carve(slab, nb, nj, ns, maxr, overlay, VELF ? 15 : 12, scan_rounds & 0xff, (scan_rounds >> 8) & 0xff);
    // Shape-generic kernels of robots with up to 12 joints (the A1) copy the row's body / joint constants into LDS once per launch
    // (behind the slab; its size arrives in the launch argument): every sub-step read them from global memory before — a wave runs
    // latency-bound, and each of those loads was an exposed trip to L2. Worth 1 - 2.5 % of the A1's 13-sub-step launch; larger
    // robots and the tuned kernels keep reading the table (their LDS decides how many envs fit a CU).
    const double *mconst = row;
    if (GENERIC && NMAX <= 18) {
        double *mc = reinterpret_cast<double *>(slab + (size_t)(scan_rounds >> 16) * 16);
        for (int i = lane; i < (int)wave_model_doubles(nb, nj); i += WV) mc[i] = row[i];
        mconst = mc;
    }
    const ModelW m{row, mconst, nb, nj, ns, SH::nb ? SH::ng : tp.n_geoms};
    // topology-only tables, built once per launch, lane-parallel (lane = body / joint): a body's first joint and joint
    // count, the joints on its chain (mask), its subtree (kids), the body every generalized coordinate
    // sits on (dbody)
    if (lane < nb) {
        L.parent[lane] = tp.body_parent[lane];
        L.jstart[lane] = 0; L.jcount[lane] = 0; L.kids[lane] = 0;
    }
    if (lane < 6) L.dbody[lane] = 0;
    for (int g = lane; g < ns; g += WV) {
        L.sbody[g] = tp.sphere_body[g];
        L.sfoot[g] = tp.sphere_foot[g];
    }
    WSYNC();
    if (lane < nj) {        // joint_body is non-decreasing: a body's joints are one run
        const int b = tp.joint_body[lane];
        L.dbody[6 + lane] = b;
        if (lane == 0 || tp.joint_body[lane - 1] != b) L.jstart[b] = lane;
        if (lane == nj - 1 || tp.joint_body[lane + 1] != b) L.jcount[b] = lane + 1;      // (the run's end for now)
    }
    WSYNC();
    if (lane < nb && L.jcount[lane] > 0) L.jcount[lane] -= L.jstart[lane];
    WSYNC();
    if (lane < nb) {
        unsigned mk = 0;
        for (int x = lane; x >= 0; x = L.parent[x]) {
            mk |= ((1u << L.jcount[x]) - 1u) << L.jstart[x];
            atomicOr(&L.kids[x], 1 << lane);      // kids[x]: the bodies of x's subtree, x included
        }
        L.mask[lane] = (int)mk;
    }
    WSYNC();
    {   // scan tables of the kinematics pass (topology only; lane = hop / joint / body)
        const int H = nb + nj;
        int hp = -1, hb = -1;
        if (lane < nb) {
            const int pb = L.parent[lane];
            if (pb >= 0) hp = L.jcount[pb] > 0 ? nb + L.jstart[pb] + L.jcount[pb] - 1 : pb;
            hb = L.jcount[lane] == 0 ? lane : -1;
        } else if (lane < H) {
            const int j = lane - nb, b = L.dbody[6 + j];
            hp = j > L.jstart[b] ? lane - 1 : b;
            hb = j == L.jstart[b] + L.jcount[b] - 1 ? b : -1;
        }
        if (lane < H) {
            L.hbody[lane] = hb;
            if (L.rh > 0) L.hanc[lane] = hp;
        }
        // nearest joint at or above a hop (walks over bodies without joints)
        auto joint_at_or_above = [&](int h) {
            while (h >= 0 && h < nb) {
                const int pb = L.parent[h];
                h = pb < 0 ? -1 : (L.jcount[pb] > 0 ? nb + L.jstart[pb] + L.jcount[pb] - 1 : pb);
            }
            return h < 0 ? -1 : h - nb;
        };
        if (lane >= nb && lane < H && L.jr > 0) L.janc[lane - nb] = joint_at_or_above(hp);
        if (lane < nb) L.bjoint[lane] = L.jcount[lane] > 0 ? L.jstart[lane] + L.jcount[lane] - 1 : joint_at_or_above(hp);
        WSYNC();
        for (int r = 1; r < L.rh; ++r) {
            if (lane < H) {
                const int a = L.hanc[(r - 1) * H + lane];
                L.hanc[r * H + lane] = a >= 0 ? L.hanc[(r - 1) * H + a] : -1;
            }
            WSYNC();
        }
        for (int r = 1; r < L.jr; ++r) {
            if (lane < nj) {
                const int a = L.janc[(r - 1) * nj + lane];
                L.janc[r * nj + lane] = a >= 0 ? L.janc[(r - 1) * nj + a] : -1;
            }
            WSYNC();
        }
    }
    if (lane < 3) {
        L.base[lane] = st.pos[(size_t)lane * n_envs + e];
        L.base[12 + lane] = st.vel[(size_t)lane * n_envs + e];
        L.base[15 + lane] = st.omega[(size_t)lane * n_envs + e];
    }
    if (lane < 9) L.base[3 + lane] = st.rot[(size_t)lane * n_envs + e];
    if (lane < nj) {
        L.q[lane] = st.q[(size_t)lane * n_envs + e];
        L.qd[lane] = st.qd[(size_t)lane * n_envs + e];
        L.tau[lane] = prm.actuation != 0 ? 0.0 : motor_torque(prm, m.motor()[lane], action[(size_t)e * nj + lane]);
    }
    WSYNC();
    unsigned long long touch[2] = {0ull, 0ull};     // proxies in contact with the ground / terrain in the last sub-step
#ifdef MG_WALKER_PROFILE
    unsigned long long ph_k0 = __builtin_readcyclecounter();
#endif
    // this env's terrain course (mg_walker_params.terrain_id: a table of courses, one per robot; else the batch's one course)
    const double *terrain = prm.terrain;
    if (GENERIC && prm.terrain_id != nullptr)
        terrain += (size_t)__builtin_amdgcn_readfirstlane(prm.terrain_id[e]) * prm.n_terrain_boxes * MG_WALKER_BOX_DOUBLES;
    // per-robot dynamics (mg_walker_params.gravity_env / foot_friction_env), else the shared values
    V3 gvec = v3(0, 0, -prm.gravity);
    double foot_mu = -1.0;
    if (GENERIC && prm.gravity_env != nullptr)
        gvec = v3(prm.gravity_env[e], prm.gravity_env[(size_t)n_envs + e], prm.gravity_env[2 * (size_t)n_envs + e]);
    if (GENERIC && prm.foot_friction_env != nullptr) foot_mu = prm.foot_friction_env[e];
    ActLane act{0.0, 0.0, 0.0, 0.0, 0.0};
    if (prm.actuation != 0 && lane < nj) {
        if (GENERIC && prm.actuation == 3) {          // HYBRID: desired angle, kp, desired rate, kd, additional torque
            const double *c5 = prm.pd_command + ((size_t)5 * lane) * n_envs + e;
            act = ActLane{c5[0], c5[(size_t)n_envs], c5[2 * (size_t)n_envs], c5[3 * (size_t)n_envs], c5[4 * (size_t)n_envs]};
        } else {
            act.q_des = prm.pd_command[(size_t)lane * n_envs + e];
            if (GENERIC) {
                act.kp = prm.pd_kp_env ? prm.pd_kp_env[(size_t)lane * n_envs + e] : prm.pd_kp[lane];
                act.kd = prm.pd_kd_env ? prm.pd_kd_env[(size_t)lane * n_envs + e] : prm.pd_kd[lane];
            }
        }
    }
    for (int it = 0; it < prm.frame_skip; ++it) {
        double *log_row = prm.substep_log ? prm.substep_log + ((size_t)it * (3 * nj + 7)) * n_envs + e : nullptr;
        wave_substep<NMAX, GENERIC, VELF>(tp, m, prm, L, lane, maxr, touch, act, log_row, n_envs,
                                    (st.foot_force != nullptr && it == prm.frame_skip - 1) ? st.foot_force + e : nullptr, nf,
                                    (GENERIC && prm.ext_wrench != nullptr && it == 0) ? prm.ext_wrench + e : nullptr, terrain, gvec, foot_mu);
    }
    if (st.bad_contacts != nullptr) {       // a1.py:314-323 GetBadFootContacts: contact points on links that are no foot
        int bad = 0;
        for (int ch = 0; ch < (GENERIC ? (ns + WV - 1) / WV : 1); ++ch) {
            const int g = ch * WV + lane;
            bad += __popcll(__ballot(g < ns && ((touch[GENERIC ? ch : 0] >> lane) & 1ull) && L.sfoot[g] < 0));
        }
        if (lane == 0) st.bad_contacts[e] = bad;
    }
    // ---- calc_state (walker_base.py:31-64) on the current configuration ------------------------------
    // after_reset = false: post-step state; the obs carries the PREVIOUS step's feet flags and the flags
    // are then refreshed from this step's contacts (walker_base_env.py:46 vs :57-63).
    // after_reset = true: first observation of a new episode; feet flags are zero (walker_base.py:18).
    auto clip5 = [](float v) { return v < -5.0f ? -5.0f : (v > 5.0f ? 5.0f : v); };
    float *ob = obs + (size_t)e * obs_dim;
    float head[8];
    auto calc_state = [&](bool after_reset, double &dist, int &at_limit, bool &all_finite) {
        wave_kinematics<false>(m, L, lane, false);
        const int pw = lane < nb ? part_weight(tp, lane) : 0;
        const double sxm = wave_sum(pw * (lane < nb ? L.o[3 * lane] : 0.0)), sym = wave_sum(pw * (lane < nb ? L.o[3 * lane + 1] : 0.0));
        const int parts = (int)wave_sum((double)pw) + (prm.floor_in_parts ? 1 : 0);
        float jp = 0.0f, jv = 0.0f;
        bool lim = false;
        if (lane < nj) {
            const double lo = m.joint_lo()[lane], hi = m.joint_hi()[lane];
            jp = (float)(2 * (L.q[lane] - 0.5 * (lo + hi)) / (hi - lo));
            jv = (float)(0.1 * L.qd[lane]);
            lim = fabsf(jp) > 0.99f;
        }
        at_limit = __popcll(__ballot(lim));
        if (lane < nj) { ob[8 + 2 * lane] = clip5(jp); ob[9 + 2 * lane] = clip5(jv); }
        bool finite = isfinite(jp) && isfinite(jv);
        // feet in contact now: one ballot per foot over the proxies (lane = proxy), not a loop over the proxies per foot
        float cnow = 0.0f;
        if (!after_reset)
            for (int f = 0; f < nf; ++f) {
                bool any = false;
                for (int ch = 0; ch < (GENERIC ? (ns + WV - 1) / WV : 1); ++ch) {
                    const int g = ch * WV + lane;
                    any = any || __ballot(g < ns && ((touch[GENERIC ? ch : 0] >> lane) & 1ull) && L.sfoot[g < ns ? g : 0] == f) != 0ull;
                }
                if (lane == f && any) cnow = 1.0f;
            }
        if (lane < nf) {
            if (after_reset) {
                ob[8 + 2 * nj + lane] = 0.0f;
                st.feet_contact[(size_t)lane * n_envs + e] = 0.0f;
            } else {
                const float prev = st.feet_contact[(size_t)lane * n_envs + e];
                ob[8 + 2 * nj + lane] = clip5(prev);
                st.feet_contact[(size_t)lane * n_envs + e] = cnow;
            }
        }
        dist = 0.0;
        // the f64 inverse-trigonometric and trigonometric calls of the head (~150-200 instructions each), spread over lanes: the
        // three atan2 run as ONE call on lanes 0-2, the two sin / cos pairs as ONE sincos on lanes 0-1 (they were eight calls on lane 0)
        const double cnt = (double)parts;
        const double bx = sxm / cnt, by = sym / cnt;
        const double *R = L.R;
        const double dx = prm.walk_target_x - bx, dy = prm.walk_target_y - by;
        double at = 0.0;
        if (lane < 3) at = atan2(lane == 0 ? R[7] : (lane == 1 ? R[3] : dy), lane == 0 ? R[8] : (lane == 1 ? R[0] : dx));
        const double roll = lane_value(at, 0), yaw = lane_value(at, 1), theta = lane_value(at, 2);
        const double ang = theta - yaw;
        double s2 = 0.0, c2 = 1.0;
        if (lane < 2) sincos(lane == 0 ? ang : -yaw, &s2, &c2);
        const double sin_ang = lane_value(s2, 0), cos_ang = lane_value(c2, 0), sn = lane_value(s2, 1), c = lane_value(c2, 1);
        if (lane == 0) {
            const double z = L.o[2];
            double sp = -R[6];
            sp = sp < -1.0 ? -1.0 : (sp > 1.0 ? 1.0 : sp);
            const double pitch = asin(sp);
            dist = sqrt(dy * dy + dx * dx);
            const double vx = c * L.base[12] - sn * L.base[13], vy = sn * L.base[12] + c * L.base[13], vz = L.base[14];
            head[0] = clip5((float)(z - prm.initial_z)); head[1] = clip5((float)sin_ang); head[2] = clip5((float)cos_ang);
            head[3] = clip5((float)(0.3 * vx)); head[4] = clip5((float)(0.3 * vy)); head[5] = clip5((float)(0.3 * vz));
            head[6] = clip5((float)roll); head[7] = clip5((float)pitch);
            for (int i = 0; i < 8; ++i) { ob[i] = head[i]; finite = finite && isfinite(head[i]); }
        }
        all_finite = __all(finite);
    };
    // One call site for calc_state (it inlines the kinematics pass; a second copy pushed the kernel into
    // scratch): pass 0 observes the stepped state and scores it, pass 1 — only for an env that ended with
    // auto_reset on — observes the freshly reset state.
    for (int pass = 0; pass < 2; ++pass) {
        double dist;
        int at_limit;
        bool all_finite;
        calc_state(pass == 1, dist, at_limit, all_finite);
        if (pass == 1) {
            if (lane == 0) {
                st.potential[e] = -dist / (prm.time_step * prm.frame_skip);
                st.steps[e] = 0;
            }
            break;
        }
        int ended = 0;
        if (lane == 0) {
            const double alive = (alive_height(prm, head[0]) > prm.alive_z) ? prm.alive_bonus : prm.dead_bonus;
            const double pot_old = st.potential[e];
            const double pot = -dist / (prm.time_step * prm.frame_skip);
            const double progress = pot - pot_old;
            const double limit_cost = prm.joints_at_limit_cost * at_limit;
            st.potential[e] = pot;
            const int steps = st.steps[e] + 1;
            st.steps[e] = steps;
            reward[e] = (float)(alive + progress + 0.0 + limit_cost + 0.0);
            if (rewards5) {
                float *r5 = rewards5 + (size_t)e * 5;
                r5[0] = (float)alive; r5[1] = (float)progress; r5[2] = 0.0f; r5[3] = (float)limit_cost; r5[4] = 0.0f;
            }
            ended = (alive < 0) || !all_finite || (steps >= prm.max_steps);
            done[e] = (uint8_t)ended;
        }
        ended = __builtin_amdgcn_readfirstlane(ended);
        if (!(prm.auto_reset && ended)) break;
        // fused auto-reset: robot_specific_reset (walker_base.py:13-24) with device-side joint noise; the
        // returned obs row is the first observation of the next episode (vector-env convention)
        WSYNC();
        if (lane < 3) {
            L.base[lane] = m.body_pos()[lane];
            L.base[12 + lane] = 0.0;
            L.base[15 + lane] = 0.0;
        }
        if (lane < 9) L.base[3 + lane] = m.body_rot()[lane];
        if (lane < nj) {
            L.q[lane] = reset_joint_noise(prm, e, lane);
            L.qd[lane] = 0.0;
        }
        WSYNC();
    }
#ifdef MG_WALKER_PROFILE
    if (lane == 0) atomicAdd(&mg_walker_phase_cycles[15], __builtin_readcyclecounter() - ph_k0);   // sub-steps + calc_state
#endif
    // ---- state store ----------------------------------------------------------------------------------
    if (lane < 3) {
        st.pos[(size_t)lane * n_envs + e] = L.base[lane];
        st.vel[(size_t)lane * n_envs + e] = L.base[12 + lane];
        st.omega[(size_t)lane * n_envs + e] = L.base[15 + lane];
    }
    if (lane < 9) st.rot[(size_t)lane * n_envs + e] = L.base[3 + lane];
    if (lane < nj) {
        st.q[(size_t)lane * n_envs + e] = L.q[lane];
        st.qd[(size_t)lane * n_envs + e] = L.qd[lane];
    }
}

inline size_t ndof_of(const mg_walker_topology *tp) { return 6 + (size_t)tp->n_joints; }

int check_walker(const mg_walker_topology *tp, const mg_walker_models *ms, const mg_walker_params *prm,
                 const mg_walker_state *st, int n) {
    if (!tp || !ms || !prm || !st) return mg::set_error(MG_ERR_NULL_POINTER, "walker: NULL descriptor");
    if (n <= 0) return mg::set_error(MG_ERR_BAD_SIZE, "n_envs=%d", n);
    if (tp->n_bodies < 1 || tp->n_bodies > NB || tp->n_joints < 0 || tp->n_joints > NJ || tp->n_spheres < 0 ||
        tp->n_spheres > NS || tp->n_feet < 0 || tp->n_feet > MG_WALKER_MAX_FEET)
        return mg::set_error(MG_ERR_BAD_SIZE, "walker topology out of range (bodies %d joints %d spheres %d feet %d)",
                             tp->n_bodies, tp->n_joints, tp->n_spheres, tp->n_feet);
    if (tp->n_geoms < 0 || tp->n_geoms > MG_WALKER_MAX_GEOMS || tp->n_pairs < 0 || tp->n_pairs > MG_WALKER_MAX_PAIRS)
        return mg::set_error(MG_ERR_BAD_SIZE, "walker topology: geoms %d pairs %d", tp->n_geoms, tp->n_pairs);
    for (int g = 0; g < tp->n_spheres; ++g)      // ABI 4 field: a zero-initialised topology would silently report every proxy to foot 0
    {
        if (tp->sphere_foot[g] < -1 || tp->sphere_foot[g] >= tp->n_feet)
            return mg::set_error(MG_ERR_BAD_CONFIG, "walker topology: sphere_foot[%d] = %d (want -1 or a foot index < %d)", g,
                                 tp->sphere_foot[g], tp->n_feet);
        if (tp->sphere_foot[g] >= 0 && tp->sphere_body[g] != tp->foot_body[tp->sphere_foot[g]])
            return mg::set_error(MG_ERR_BAD_CONFIG, "walker topology: proxy %d sits on body %d but reports to foot %d on body %d "
                                 "(sphere_foot is new in ABI 4: -1 = no foot, else a foot whose body carries the proxy)", g,
                                 tp->sphere_body[g], tp->sphere_foot[g], tp->foot_body[tp->sphere_foot[g]]);
    }
    const int need = 25 * tp->n_bodies + 12 * tp->n_joints + 4 * tp->n_spheres + 7 * tp->n_geoms +
                     (prm->sphere_margin_in_table ? tp->n_spheres : 0);
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

#ifdef MG_WALKER_PROFILE
extern "C" int mg_walker_profile_read(unsigned long long *out16, int clear) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(mg_walker_phase_cycles), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (clear) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(mg_walker_phase_cycles), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

extern "C" int mg_walker_reset(const mg_walker_topology *tp, const mg_walker_models *ms, const mg_walker_params *prm,
                               int32_t n, const mg_walker_state *st, const uint8_t *mask, const double *joint_noise,
                               float *obs, void *stream) {
    if (int rc = check_walker(tp, ms, prm, st, n)) return rc;
    mg::DeviceGuard guard(mg::device_of(st->pos));
    hipLaunchKernelGGL(walker_reset_kernel, dim3((n + WK_BLOCK - 1) / WK_BLOCK), dim3(WK_BLOCK), 0, (hipStream_t)stream,
                       *tp, *ms, *prm, *st, n, mask, joint_noise, obs);
    return mg::check_launch("walker_reset_kernel");
}

extern "C" int mg_walker_step(const mg_walker_topology *tp, const mg_walker_models *ms, const mg_walker_params *prm,
                              int32_t n, const mg_walker_state *st, const float *action, float *obs, float *reward,
                              float *rewards5, uint8_t *done, void *stream) {
    if (int rc = check_walker(tp, ms, prm, st, n)) return rc;
    if (prm->actuation == 0) MG_REQUIRE_PTR(action);
    else {
        if (prm->actuation < 1 || prm->actuation > 4) return mg::set_error(MG_ERR_BAD_CONFIG, "walker actuation %d", prm->actuation);
        if (prm->mapping == 0) return mg::set_error(MG_ERR_UNSUPPORTED, "in-launch actuators need the wave mapping");
        MG_REQUIRE_PTR(prm->pd_command);
    }
    if (prm->n_terrain_boxes < 0) return mg::set_error(MG_ERR_BAD_SIZE, "walker terrain: %d boxes", prm->n_terrain_boxes);
    if (prm->n_terrain_boxes > 0) {
        if (prm->mapping == 0) return mg::set_error(MG_ERR_UNSUPPORTED, "terrain boxes need the wave mapping");
        MG_REQUIRE_PTR(prm->terrain);
        if (prm->terrain_id != nullptr && prm->n_terrain_tables < 1)
            return mg::set_error(MG_ERR_BAD_SIZE, "walker terrain table: %d courses", prm->n_terrain_tables);
    }
    MG_REQUIRE_PTR(obs);
    MG_REQUIRE_PTR(reward);
    MG_REQUIRE_PTR(done);
    mg::DeviceGuard guard(mg::device_of(st->pos));
    if (prm->mapping == 0) {   // lane-per-env reference mapping (private-memory work set)
        if (tp->n_spheres > 64 || prm->sphere_friction != nullptr || st->foot_force != nullptr || prm->ext_wrench != nullptr ||
            prm->gravity_env != nullptr)
            return mg::set_error(MG_ERR_UNSUPPORTED, "per-proxy friction, foot forces, pushes, per-robot gravity and > 64 collision proxies need the wave mapping");
        hipLaunchKernelGGL(walker_step_kernel, dim3((n + WK_BLOCK - 1) / WK_BLOCK), dim3(WK_BLOCK), 0,
                           (hipStream_t)stream, *tp, *ms, *prm, *st, n, action, obs, reward, rewards5, done);
        return mg::check_launch("walker_step_kernel");
    }
    if (tp->n_spheres > 128 || 6 + tp->n_joints > 64)
        return mg::set_error(MG_ERR_BAD_SIZE, "wave mapping needs <= 128 collision proxies and <= 58 joints");
    const int maxr = 3 * W_MAXC + tp->n_joints;
    using Humanoid = Shape<13, 17, 29, 17, 1>;
    using Ant = Shape<13, 8, 25, 13, 1>;
    using HumanoidDamped = Shape<13, 17, 29, 17, 1, 1>;     // + body velocity damping (the "bullet" preset, the envs' default)
    using AntDamped = Shape<13, 8, 25, 13, 1, 1>;
    // (terrain boxes, per-proxy friction and > 64 proxies are compiled into the shape-generic instantiations only: the
    // tuned kernels keep their registers)
    const bool damped = prm->body_linear_damping != 0.0 || prm->body_angular_damping != 0.0;
    if (prm->foot_friction_env != nullptr && prm->sphere_friction == nullptr)
        return mg::set_error(MG_ERR_BAD_CONFIG, "walker: foot_friction_env needs sphere_friction (per-proxy friction)");
    const bool generic_only = prm->n_terrain_boxes != 0 || prm->sphere_friction != nullptr || prm->gravity_env != nullptr ||
                              st->foot_force != nullptr || prm->actuation == 3 ||
                              prm->pd_kp_env != nullptr || prm->pd_kd_env != nullptr || prm->ext_wrench != nullptr;
    auto shape_is = [&](int b, int j, int s, int g) {
        return !generic_only && tp->n_bodies == b && tp->n_joints == j && tp->n_spheres == s && tp->n_geoms == g;
    };
    bool tuned = shape_is(Humanoid::nb, Humanoid::nj, Humanoid::ns, Humanoid::ng) || shape_is(Ant::nb, Ant::nj, Ant::ns, Ant::ng);
    int fd = (tuned && !damped) ? 12 : 15;           // velocity-product frame doubles per body (wave_lds_doubles)
    // scratch use of the Jh block during the M / h assembly: the composite-rigid-body tables from the front and, when
    // they also fit, the 12 doubles per body of velocity-product frames from the back
    bool overlay = false;
    {
        const size_t block = (size_t)maxr * ndof_of(tp), need = wave_assembly_doubles(tp->n_bodies, tp->n_joints);
        if (need > block)
            return mg::set_error(MG_ERR_UNSUPPORTED, "walker topology needs %zu scratch doubles for the mass-matrix "
                                 "assembly, the wave mapping has %zu: use mapping = lane", need, block);
        overlay = need + fd * (size_t)tp->n_bodies <= block;
        if (tuned && !overlay) {        // (cannot happen for the two shipped shapes; keeps host and kernel layouts in step)
            tuned = false;
            fd = 15;
            overlay = need + fd * (size_t)tp->n_bodies <= block;
        }
    }
    {   // the detection pass records its contact candidates in the Jh block behind the capsules' end points (9 doubles each):
        // either W_MAXCAND of them fit, or every candidate this topology can produce does — the oracle's cap is then never reached
        const size_t block = (size_t)maxr * ndof_of(tp), seg = 8 * (size_t)tp->n_geoms;
        const size_t cap = block > seg ? (block - seg) / 9 : 0;
        const size_t possible = (size_t)tp->n_spheres * (prm->n_terrain_boxes > 0 ? 2 : 1) + (prm->self_collision ? (size_t)tp->n_pairs : 0);
        if (cap < (size_t)W_MAXCAND && cap < possible)
            return mg::set_error(MG_ERR_UNSUPPORTED, "walker topology: %zu contact candidates possible, the wave mapping has scratch for %zu "
                                 "(< %d): use mapping = lane", possible, cap, W_MAXCAND);
    }
    // rounds of the kinematics scans: 2^rh >= longest chain of hops (body offsets + joints), 2^jr >= longest chain of joints
    int rh = 0, jr = 0;
    {
        int hops[NB], joints[NB], max_h = 0, max_j = 0, j = 0;      // per body: hops / joints on the chain from the base, inclusive
        for (int b = 0; b < tp->n_bodies; ++b) {
            const int pb = tp->body_parent[b];
            if (pb >= b) return mg::set_error(MG_ERR_BAD_CONFIG, "walker topology: body %d has parent %d (parents come first)", b, pb);
            int cnt = 0;
            while (j < tp->n_joints && tp->joint_body[j] == b) { ++j; ++cnt; }
            hops[b] = (pb < 0 ? 0 : hops[pb]) + 1 + cnt;
            joints[b] = (pb < 0 ? 0 : joints[pb]) + cnt;
            max_h = hops[b] > max_h ? hops[b] : max_h;
            max_j = joints[b] > max_j ? joints[b] : max_j;
        }
        while ((1 << rh) < max_h) ++rh;
        while ((1 << jr) < max_j) ++jr;
        if (tp->n_joints > 0 && jr < 1) jr = 1;     // round 0 of the joint table is also the "previous joint" table: always there
        if (wave_scan_doubles(tp->n_bodies, tp->n_joints) + (overlay ? fd * (size_t)tp->n_bodies : 0) > (size_t)maxr * ndof_of(tp))
            return mg::set_error(MG_ERR_UNSUPPORTED, "walker topology: the kinematics scan needs %zu scratch doubles, the wave "
                                 "mapping has %zu: use mapping = lane", wave_scan_doubles(tp->n_bodies, tp->n_joints),
                                 (size_t)maxr * ndof_of(tp) - (overlay ? fd * (size_t)tp->n_bodies : 0));
    }
    size_t lds = wave_lds_doubles(tp->n_bodies, tp->n_joints, maxr, overlay, fd) * sizeof(double) +
                 wave_lds_ints(tp->n_bodies, tp->n_joints, tp->n_spheres, maxr, rh, jr) * sizeof(int);
    lds = (lds + 15) & ~size_t(15);
    // shape-generic kernels of robots with <= 12 joints keep the body and joint constants of the env's model row in LDS for the whole
    // launch (wave_model_doubles; the instantiations with NMAX <= 18)
    const size_t lds_slab = lds;
    if (!tuned && 6 + tp->n_joints <= 18) lds += wave_model_doubles(tp->n_bodies, tp->n_joints) * sizeof(double);
#ifdef MG_WALKER_LDS_FLOOR      /* timing experiment only (scripts/walker_occupancy_probe.py): fewer resident envs per CU */
    if (const char *fl = getenv("MG_WALKER_LDS_FLOOR")) { const size_t f = (size_t)atol(fl); if (f > lds && f <= 64 * 1024) lds = f; }
#endif
    const int maxr_flags = overlay ? -maxr : maxr;
    if (lds > 160 * 1024) return mg::set_error(MG_ERR_BAD_SIZE, "walker needs %zu B of LDS", lds);
    if (lds > 64 * 1024) return mg::set_error(MG_ERR_BAD_SIZE, "walker needs %zu B of LDS (> 64 KiB)", lds);
    const int ndof = 6 + tp->n_joints;
    // the substitution keeps its vector in registers, so the dof count is a template parameter:
    // 14 = ant, 18 = quadruped, 23 = humanoid, 30 = the ABI maximum
    // ... and so is the whole robot shape for the two robots MetaLocomotion ships (LDS addresses and model-table
    // offsets become literals); any other topology runs the shape-generic instantiations
    auto is_shape = [&](int b, int j, int s, int g) { return tuned && shape_is(b, j, s, g); };
#define MG_WALKER_LAUNCH(NMAX_, SHAPE_)                                                                                  \
    hipLaunchKernelGGL((walker_step_wave_kernel<NMAX_, SHAPE_>), dim3(n), dim3(WV), lds, (hipStream_t)stream, *tp, *ms, \
                       *prm, *st, n, maxr_flags, rh | (jr << 8) | ((int)(lds_slab / 16) << 16), action, obs, reward, rewards5, done)
    if (is_shape(Humanoid::nb, Humanoid::nj, Humanoid::ns, Humanoid::ng)) {
        if (damped) MG_WALKER_LAUNCH(23, HumanoidDamped);
        else MG_WALKER_LAUNCH(23, Humanoid);
    } else if (is_shape(Ant::nb, Ant::nj, Ant::ns, Ant::ng)) {
        if (damped) MG_WALKER_LAUNCH(14, AntDamped);
        else MG_WALKER_LAUNCH(14, Ant);
    }
    else if (ndof == 14) MG_WALKER_LAUNCH(14, ShapeDof<8>);
    else if (ndof < 14) MG_WALKER_LAUNCH(14, ShapeAny);
    else if (ndof == 18) MG_WALKER_LAUNCH(18, ShapeDof<12>);  // a quadruped: 6 + 12 (the A1)
    else if (ndof < 18) MG_WALKER_LAUNCH(18, ShapeAny);
    else if (ndof == 23) MG_WALKER_LAUNCH(23, ShapeDof<17>);
    else if (ndof < 23) MG_WALKER_LAUNCH(23, ShapeAny);
    else MG_WALKER_LAUNCH(ND, ShapeAny);                      // (a 30-dof exact instantiation spills VGPRs)
#undef MG_WALKER_LAUNCH
    return mg::check_launch("walker_step_wave_kernel");
}
