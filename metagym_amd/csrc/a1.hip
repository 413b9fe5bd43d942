// a1.hip — Quadrupedal (Unitree A1) actuation path for gfx950: what Minitaur._StepInternal
// (metagym/quadrupedal/robots/minitaur.py:232-238) does around pybullet.stepSimulation(), for N robots per launch.
//
//   mg_a1_apply_action         A1.ApplyAction a1.py:451-483, Minitaur.ApplyAction minitaur.py:906-955,
//                              ProcessAction :1419-1436, _GetPDObservation / _GetDelayedObservation :1205-1232,
//                              LaikagoMotorModel.convert_to_torque laikago_motor.py:92-169
//   mg_a1_receive_observation  ReceiveObservation minitaur.py:1184-1203, GetTrueObservation :1175-1182
//   mg_a1_sensors              GetMotorAngles / Velocities / Torques :755-810, GetBaseRollPitchYawRate :874-885,
//                              GetEnergyConsumptionPerControlStep :812-820
//
// The A1 body is not part of this file (a1.urdf and PyBullet are not in the reference tree). Everything is float64
// element-wise arithmetic in the order NumPy evaluates it, so the results are bit-identical to the reference
// (tests/golden/a1_actuation.npz); the library is built with -ffp-contract=off.
//
// Mapping: SoA arrays ([component][N]) so a wave's loads and stores are contiguous; four lanes per robot in the per-sub-step
// kernels (see act_lane), one lane per robot elsewhere. The path is HBM-bound: per sub-step a robot writes one 43-double
// observation into its history ring and reads two ring entries for the control observation (+ two 24-double halves when
// the PD latency is not zero).
#include "mg_common.h"

namespace {

constexpr int A1_BLOCK = 64;   // one wave per workgroup: 16 384 robots still reach every CU
constexpr int NM = MG_A1_NUM_MOTORS;
constexpr int OD = MG_A1_OBS_DIM;

struct A1K {   // mg_a1_actuator_config by value (kernarg -> SGPRs)
    mg_a1_actuator_config c;
};

// _GetDelayedObservation (minitaur.py:1205-1226) for robot e: which ring entries to blend and how.
//   latency <= 0 or one entry     -> the newest
//   n = int(latency / dt); n + 1 >= len(history) -> the OLDEST entry (history[-1])
//   else (1 - a) * history[n] + a * history[n + 1],  a = (latency - n * dt) / dt
struct Delay {
    int s0, s1;      // ring slots
    double a;        // blend weight of s1; < 0: no blend, take s0 as it is
};
__device__ __forceinline__ Delay delayed(double latency, double dt, int count, int head, int hist_len) {
    Delay d;
    auto slot = [&](int k) { int s = head - k; return s < 0 ? s + hist_len : s; };
    d.s0 = slot(0); d.s1 = d.s0; d.a = -1.0;
    if (latency <= 0.0 || count == 1) return d;
    const int n_ago = (int)(latency / dt);
    if (n_ago + 1 >= count) { d.s0 = slot(count - 1); d.s1 = d.s0; return d; }
    const double remaining = latency - (double)n_ago * dt;
    d.a = remaining / dt;
    d.s0 = slot(n_ago);
    d.s1 = slot(n_ago + 1);
    return d;
}
__device__ __forceinline__ double blend(const Delay &d, const double *hist, size_t entry_stride, int comp, int n, int e) {
    const double v0 = hist[(size_t)d.s0 * entry_stride + (size_t)comp * n + e];
    if (d.a < 0.0) return v0;
    const double v1 = hist[(size_t)d.s1 * entry_stride + (size_t)comp * n + e];
    return (1.0 - d.a) * v0 + d.a * v1;
}

// MapToMinusPiToPi minitaur.py:56-73
__device__ __forceinline__ double map_to_pi(double x) {
    const double two_pi = 2.0 * 3.141592653589793;
    double m = fmod(x, two_pi);
    if (m >= 3.141592653589793) m -= two_pi;
    else if (m < -3.141592653589793) m += two_pi;
    return m;
}

// Lane mapping of the actuation kernels: FOUR lanes per robot, lane `sub` owning motors 3 sub .. 3 sub + 2 (their angle,
// velocity and torque components) and its share of the 7 base components — 16 robots per wave, a component's 16 values
// still contiguous. One lane per robot left 16 384 robots on 256 waves (a quarter of the chip's SIMDs, one dependent chain
// of ~180 memory operations each); per-motor arithmetic is independent, so the results are bit-identical.
constexpr int ACT_SUBS = 4, ACT_MOTORS = NM / ACT_SUBS, ACT_ROBOTS_PER_WAVE = 64 / ACT_SUBS;
__device__ __forceinline__ bool act_lane(int n, int &e, int &sub) {
    const int gid = blockIdx.x * A1_BLOCK + threadIdx.x, lane = gid & 63;
    e = (gid >> 6) * ACT_ROBOTS_PER_WAVE + (lane & (ACT_ROBOTS_PER_WAVE - 1));
    sub = lane / ACT_ROBOTS_PER_WAVE;
    return e < n;
}
__host__ inline unsigned act_blocks(int n) {
    const long waves = ((long)n + ACT_ROBOTS_PER_WAVE - 1) / ACT_ROBOTS_PER_WAVE;
    return (unsigned)((waves * 64 + A1_BLOCK - 1) / A1_BLOCK);
}
struct Ring { int count, head; };

__device__ __forceinline__ void a1_apply(const mg_a1_actuator_config &c, const mg_a1_actuator_state &st, int n, int e, int sub,
                                         Ring ring, const double *command, const double *last_command, double lerp,
                                         double *torque) {
    const size_t stride = (size_t)OD * n;
    const double pd_lat = c.pd_latency_env ? c.pd_latency_env[e] : c.pd_latency;
    const Delay d = delayed(pd_lat, c.time_step, ring.count, ring.head, c.history_len);
    auto cmd = [&](int i) {      // ProcessAction: last + lerp * (action - last)
        const double a = command[(size_t)i * n + e];
        if (last_command == nullptr) return a;
        const double l = last_command[(size_t)i * n + e];
        return l + lerp * (a - l);
    };
#pragma unroll
    for (int m = 0; m < ACT_MOTORS; ++m) {
        const int i = ACT_MOTORS * sub + m;
        double t;
        if (c.mode == MG_A1_MODE_TORQUE) {
            t = c.strength[i] * cmd(i);                                             // laikago_motor.py:125-128
        } else {
            const double q = blend(d, st.history, stride, i, n, e);                 // _GetPDObservation :1228-1232
            const double qd = blend(d, st.history, stride, NM + i, n, e);
            double kp, kd, q_des, qd_des, extra;
            if (c.mode == MG_A1_MODE_POSITION) {
                kp = c.kp_env ? c.kp_env[(size_t)i * n + e] : c.kp[i];
                kd = c.kd_env ? c.kd_env[(size_t)i * n + e] : c.kd[i];
                q_des = cmd(i);
                if (c.clip_commands) {                                              // a1.py:465-483
                    const double cur = map_to_pi(st.control_obs[(size_t)i * n + e]);
                    const double lo = cur - c.max_angle_change, hi = cur + c.max_angle_change;
                    q_des = fmin(fmax(q_des, lo), hi);                              // np.clip = minimum(maximum(x, lo), hi)
                }
                qd_des = 0.0;
                extra = 0.0;
            } else {                                                                // HYBRID :143-153
                q_des = cmd(5 * i);
                kp = cmd(5 * i + 1);
                qd_des = cmd(5 * i + 2);
                kd = cmd(5 * i + 3);
                extra = cmd(5 * i + 4);
            }
            // -1 * (kp * (q - q_des)) - kd * (qd - qd_des) + additional_torques        :157-158
            t = (-1.0 * (kp * (q - q_des)) - kd * (qd - qd_des)) + extra;
            t = c.strength[i] * t;                                                  // :159
            if (c.has_torque_limit) {                                               // :163-168
                const double lim = c.torque_limit[i];
                t = fmin(fmax(t, -1.0 * lim), lim);
            }
        }
        st.observed_torque[(size_t)i * n + e] = t;                                  // minitaur.py:930
        torque[(size_t)i * n + e] = t * 1.0;                                        // motor_direction = 1 (a1.py:43)
    }
}

// The history components lane `sub` owns: 3 x (angle, velocity, torque) + base component sub (and sub + 4 when < 7).
constexpr int ACT_COMPS = 3 * ACT_MOTORS + 2;
__device__ __forceinline__ int act_comp(int sub, int k) {      // k-th owned component, or -1
    if (k < 3 * ACT_MOTORS) return (k / ACT_MOTORS) * NM + ACT_MOTORS * sub + k % ACT_MOTORS;
    const int x = sub + ACT_SUBS * (k - 3 * ACT_MOTORS);
    return x < OD - 3 * NM ? 3 * NM + x : -1;
}

__device__ __forceinline__ Ring a1_receive(const mg_a1_actuator_config &c, const mg_a1_actuator_state &st, int n, int e, int sub,
                                           const double *q, const double *qd, const double *quat, const double *rate,
                                           const uint8_t *clear_mask) {
    const size_t stride = (size_t)OD * n;
    int count = st.count[e], head = st.head[e];
    const int cm = clear_mask != nullptr ? clear_mask[e] : 0;
    if (cm == 2) return Ring{count, head};                                          // not this robot's call (partial reset)
    if (cm == 1) count = 0;                                                         // _observation_history.clear()
    head = count == 0 ? 0 : (head + 1 == c.history_len ? 0 : head + 1);             // appendleft
    if (count < c.history_len) ++count;
    double *slot = st.history + (size_t)head * stride + e;
    double in[ACT_COMPS];                                                           // (all loads in flight, then the stores)
#pragma unroll
    for (int k = 0; k < ACT_COMPS; ++k) {
        double v = 0.0;
        if (k < 3 * ACT_MOTORS) {      // (k is a literal after unrolling: the group is decided at compile time)
            const int i = ACT_MOTORS * sub + k % ACT_MOTORS;
            // GetTrueMotorAngles (:743-753): (angle - offset) * direction with offset 0, direction 1 (a1.py:43-49)
            if (k / ACT_MOTORS == 0) v = (q[(size_t)i * n + e] - 0.0) * 1.0;
            else if (k / ACT_MOTORS == 1) v = qd[(size_t)i * n + e] * 1.0;
            else v = st.observed_torque[(size_t)i * n + e];
        } else {
            const int x = sub + ACT_SUBS * (k - 3 * ACT_MOTORS);      // base quaternion (4) then body rates (3)
            if (x < 4) v = quat[(size_t)x * n + e];
            else if (x < 7) v = rate[(size_t)(x - 4) * n + e];
        }
        in[k] = v;
    }
#pragma unroll
    for (int k = 0; k < ACT_COMPS; ++k) {
        const int comp = act_comp(sub, k);
        if (comp >= 0) slot[(size_t)comp * n] = in[k];
    }
    // the four lanes of a robot sit in one wave and all of them loaded count / head in the instruction above: one writes
    if (sub == 0) { st.count[e] = count; st.head[e] = head; }
    // a lane reads back only what it wrote itself (its own components of the ring): no fence needed
    const double lat = c.control_latency_env ? c.control_latency_env[e] : c.control_latency;
    const Delay d = delayed(lat, c.time_step, count, head, c.history_len);
    // all loads first, then the stores: written as load -> store per component the compiler must assume the
    // control-observation store may alias the history and serialises the round trips
    double v[ACT_COMPS];
#pragma unroll
    for (int k = 0; k < ACT_COMPS; ++k) {
        const int comp = act_comp(sub, k);
        v[k] = comp >= 0 ? blend(d, st.history, stride, comp, n, e) : 0.0;
    }
#pragma unroll
    for (int k = 0; k < ACT_COMPS; ++k) {
        const int comp = act_comp(sub, k);
        if (comp >= 0) st.control_obs[(size_t)comp * n + e] = v[k];
    }
    return Ring{count, head};
}

__global__ __launch_bounds__(A1_BLOCK) void a1_apply_action_kernel(A1K k, mg_a1_actuator_state st, int n,
                                                                   const double *command, const double *last_command,
                                                                   double lerp, double *torque) {
    int e, sub;
    if (act_lane(n, e, sub)) a1_apply(k.c, st, n, e, sub, Ring{st.count[e], st.head[e]}, command, last_command, lerp, torque);
}

__global__ __launch_bounds__(A1_BLOCK) void a1_receive_kernel(A1K k, mg_a1_actuator_state st, int n, const double *q,
                                                              const double *qd, const double *quat, const double *rate,
                                                              const uint8_t *clear_mask) {
    int e, sub;
    if (act_lane(n, e, sub)) a1_receive(k.c, st, n, e, sub, q, qd, quat, rate, clear_mask);
}

// ReceiveObservation of sub-step k and ApplyAction of sub-step k + 1 in one launch. A lane only ever reads ring entries it
// wrote itself, in program order (the PD observation of a motor is made of that motor's own components), so no fence is
// needed between the two halves; the ring position travels in registers.
__global__ __launch_bounds__(A1_BLOCK) void a1_receive_apply_kernel(A1K k, mg_a1_actuator_state st, int n, const double *q,
                                                                    const double *qd, const double *quat, const double *rate,
                                                                    const double *command, const double *last_command,
                                                                    double lerp, double *torque) {
    int e, sub;
    if (!act_lane(n, e, sub)) return;
    const Ring ring = a1_receive(k.c, st, n, e, sub, q, qd, quat, rate, nullptr);
    a1_apply(k.c, st, n, e, sub, ring, command, last_command, lerp, torque);
}

__global__ __launch_bounds__(A1_BLOCK) void a1_sensors_kernel(A1K k, mg_a1_actuator_state st, int n, double *angles,
                                                              double *vels, double *torques, double *rate,
                                                              double *energy) {
    const int e = blockIdx.x * A1_BLOCK + threadIdx.x;
    if (e >= n) return;
    const mg_a1_actuator_config &c = k.c;
    double dot = 0.0;
#pragma unroll
    for (int i = 0; i < NM; ++i) {
        const double a = st.control_obs[(size_t)i * n + e], v = st.control_obs[(size_t)(NM + i) * n + e];
        const double t = st.control_obs[(size_t)(2 * NM + i) * n + e];
        if (angles) angles[(size_t)i * n + e] = map_to_pi(a);
        if (vels) vels[(size_t)i * n + e] = v;
        if (torques) torques[(size_t)i * n + e] = t;
        dot += t * v;      // np.dot over 12 doubles: sequential here; OpenBLAS ddot may associate differently (<= 1e-15 rel.)
    }
    if (rate)
        for (int i = 0; i < 3; ++i) rate[(size_t)i * n + e] = st.control_obs[(size_t)(3 * NM + 4 + i) * n + e];
    if (energy) energy[e] = fabs(dot) * c.time_step * (double)c.action_repeat;
}

// ---- ETGWrapper + trajectory generator (MonitorEnv.py:222-273) -----------------------------------------------------

struct EtgK { mg_a1_etg_config c; };

// foot_position_in_hip_frame_to_joint_angle robots/a1.py:88-102
__device__ __forceinline__ void leg_ik(double x, double y, double z, double l_hip_sign, double out[3]) {
    const double l_up = 0.2, l_low = 0.2, l_hip = 0.08505 * l_hip_sign;
    const double theta_knee = -acos((x * x + y * y + z * z - l_hip * l_hip - l_low * l_low - l_up * l_up) / (2 * l_low * l_up));
    const double l = sqrt(l_up * l_up + l_low * l_low + 2 * l_up * l_low * cos(theta_knee));
    const double theta_hip = asin(-x / l) - theta_knee / 2;
    const double ch = cos(theta_hip + theta_knee / 2);
    const double c1 = l_hip * y - l * ch * z;
    const double s1 = l * ch * y + l_hip * z;
    out[0] = atan2(s1, c1);
    out[1] = theta_hip;
    out[2] = theta_knee;
}

__global__ __launch_bounds__(A1_BLOCK) void a1_etg_kernel(EtgK k, int n, double *last_act, const double *action,
                                                          const double *t_in, double *command, double *etg_obs) {
    const int e = blockIdx.x * A1_BLOCK + threadIdx.x;
    if (e >= n) return;
    const mg_a1_etg_config &c = k.c;
    double total[NM];
#pragma unroll
    for (int i = 0; i < NM; ++i) total[i] = action ? action[(size_t)i * n + e] : 0.0;
    if (c.enabled) {
        if (action)                                                              // MonitorEnv.py:263
#pragma unroll
            for (int i = 0; i < NM; ++i) total[i] = total[i] + last_act[(size_t)i * n + e];
        const double t = t_in ? t_in[e] : 0.0;
        // ETG_layer.update2 ETG_model.py:38-55: x = forward(t), x2 = forward(t + T2_ratio * T)
        const double t2 = t + c.T2_ratio * c.T;
        const double x0 = c.amp * sin(c.phase[0] + t * c.omega), x1 = c.amp * sin(c.phase[1] + t * c.omega);
        const double y0 = c.amp * sin(c.phase[0] + t2 * c.omega), y1 = c.amp * sin(c.phase[1] + t2 * c.omega);
        double act1[3] = {0, 0, 0}, act2[3] = {0, 0, 0};
        for (int h = 0; h < c.H; ++h) {
            const double d0 = x0 - c.u[h][0], d1 = x1 - c.u[h][1], g0 = y0 - c.u[h][0], g1 = y1 - c.u[h][1];
            const double r = exp(-((d0 * d0 + d1 * d1) / c.sigma_sq)), r2 = exp(-((g0 * g0 + g1 * g1) / c.sigma_sq));
            if (etg_obs) etg_obs[(size_t)h * n + e] = r;
#pragma unroll
            for (int a = 0; a < 3; ++a) { act1[a] += c.w[a][h] * r; act2[a] += c.w[a][h] * r2; }      // ETG_model.forward :99-103
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) { act1[a] += c.b[a]; act2[a] += c.b[a]; }
        double ref[NM];
#pragma unroll
        for (int a = 0; a < 3; ++a) {                                            // :104-115
            ref[a] = act1[a];
            ref[3 + a] = c.gallop ? act1[a] : act2[a];
            ref[6 + a] = act2[a];
            ref[9 + a] = c.gallop ? act2[a] : act1[a];
        }
        const double base_foot[NM] = {0.18, -0.15, -0.23, 0.18, 0.148, -0.23, -0.18, -0.14, -0.23, -0.18, 0.135, -0.23};   // :5-6
        const double com[3] = {-0.012731, -0.002186, -0.000515};                 // a1.py:60
        const double hipx[4] = {0.183, 0.183, -0.183, -0.183}, hipy[4] = {-0.047, 0.047, -0.047, 0.047};               // a1.py:61-63
        const double pose_ori[3] = {0.0, 0.9, -1.8};                             // ETG_model.py:83
        for (int leg = 0; leg < 4; ++leg) {
            double ang[3];
            if (c.act_mode_pose) {                                               // act_clip :118-120
                const double sc[3] = {0.1, 0.7, 0.7};
                for (int a = 0; a < 3; ++a) ang[a] = tanh(ref[3 * leg + a]) * sc[a];
            } else {                                                             // :121-130
                double d[3] = {ref[3 * leg], ref[3 * leg + 1], ref[3 * leg + 2]};
                const double sgn = (leg & 1) ? 1.0 : -1.0;                       // (-1) ** (leg + 1)
                for (int it = 0; it < 4096; ++it) {                              // `while(1)`: |delta| shrinks 5 % per retry
                    leg_ik((d[0] + base_foot[3 * leg]) - (hipx[leg] + com[0]), (d[1] + base_foot[3 * leg + 1]) - (hipy[leg] + com[1]),
                           (d[2] + base_foot[3 * leg + 2]) - (0.0 + com[2]), sgn, ang);
                    if (!(isnan(ang[0]) || isnan(ang[1]) || isnan(ang[2]))) break;
                    d[0] *= 0.95; d[1] *= 0.95; d[2] *= 0.95;
                }
                for (int a = 0; a < 3; ++a) ang[a] = ang[a] - pose_ori[a];
            }
            for (int a = 0; a < 3; ++a) last_act[(size_t)(3 * leg + a) * n + e] = ang[a] * c.etg_weight;   // MonitorEnv.py:269
        }
    }
    if (action == nullptr) return;                                               // ETGWrapper.reset: no command
    // LaikagoPoseOffsetGenerator.get_action simple_openloop.py:144-165
#pragma unroll
    for (int i = 0; i < NM; ++i) {
        double a = total[i];
        if (c.action_space >= 2) {
            double extra = 0.0;                                                  // new_action = zeros; [6:9] = a[3:6]; [9:12] = a[:3]
            if (i >= 6 && i < 9) extra = total[i - 3];
            else if (i >= 9) extra = total[i - 9];
            a = extra + a;                                                       // new_action += input_action
        }
        command[(size_t)i * n + e] = c.pose[i] + a;
    }
}

// ---- RewardShaping (MonitorEnv.py:275-519) -------------------------------------------------------------------------

struct RewK { mg_a1_reward_config c; };

__device__ __forceinline__ void env_vec(const mg_a1_reward_config &c, int e, double posex, int &up, int &down, double &ang) {
    up = 0; down = 0; ang = 0.0;
    if (c.terrain_id != nullptr) {       // per-robot terrains: this robot's course of the segment table
        const int t = c.terrain_id[e];
        const double *seg = c.seg_table + (size_t)t * MG_A1_MAX_SEGMENTS * 5;
        for (int s = 0; s < c.seg_count[t]; ++s)
            if (posex + 0.2 >= seg[5 * s] && posex + 0.2 <= seg[5 * s + 1]) {
                up = seg[5 * s + 2] != 0.0; down = seg[5 * s + 3] != 0.0; ang = seg[5 * s + 4];
                return;
            }
        return;
    }
    for (int s = 0; s < c.n_segments; ++s)
        if (posex + 0.2 >= c.seg[s][0] && posex + 0.2 <= c.seg[s][1]) {          // :328-333
            up = c.seg[s][2] != 0.0; down = c.seg[s][3] != 0.0; ang = c.seg[s][4];
            return;
        }
}
__device__ __forceinline__ double c_prec(double v, double t, double w) {          // :421-425
    const double x = (v - t) * w;
    return tanh(x * x);
}
__device__ __forceinline__ double re_rot(const mg_a1_reward_config &c, double yaw, double d_yaw, double r) {   // :411-419
    const double two_pi = 2 * 3.141592653589793;
    const double k1 = 1 - c_prec(yaw, d_yaw, c.cw_half), k2 = 1 - c_prec(yaw, d_yaw + two_pi, c.cw_half),
                 k3 = 1 - c_prec(yaw, d_yaw - two_pi, c.cw_half);
    const double k = fmax(fmax(k1, k2), k3);
    return fmin(k * r, r);
}
// the direction block shared by re_torso (:481-497) and re_feet (:431-446); vd2 persists (mutable default argument)
__device__ __forceinline__ void direction(const mg_a1_reward_config &c, int e, double d_yaw, double posex, double &vd0, double &vd1,
                                          double &vd2) {
    vd0 = cos(d_yaw); vd1 = sin(d_yaw);
    int up, down; double ang;
    env_vec(c, e, posex, up, down, ang);
    if (up) { vd0 *= fabs(cos(ang)); vd1 *= fabs(cos(ang)); vd2 = fabs(sin(ang)); }
    else if (down) { vd0 *= fabs(cos(ang)); vd1 *= fabs(cos(ang)); vd2 = -fabs(sin(ang)); }
}
// get_foot_world :458-473: rot_mat (3 x 3) . foot^T + base
__device__ __forceinline__ void foot_world(const double *rot, const double *base, const double *foot, int n, int e, double fw[12]) {
    double R[9], b[3];
    for (int i = 0; i < 9; ++i) R[i] = rot[(size_t)i * n + e];
    for (int i = 0; i < 3; ++i) b[i] = base[(size_t)i * n + e];
    for (int f = 0; f < 4; ++f) {
        const double x = foot[(size_t)(3 * f) * n + e], y = foot[(size_t)(3 * f + 1) * n + e], z = foot[(size_t)(3 * f + 2) * n + e];
        for (int r = 0; r < 3; ++r) fw[3 * f + r] = ((R[3 * r] * x + R[3 * r + 1] * y) + R[3 * r + 2] * z) + b[r];
    }
}

__global__ __launch_bounds__(A1_BLOCK) void a1_reward_reset_kernel(mg_a1_reward_state st, int n, const double *base,
                                                                   const double *rot, const double *foot, const uint8_t *mask) {
    const int e = blockIdx.x * A1_BLOCK + threadIdx.x;
    if (e >= n || (mask && !mask[e])) return;
    double fw[12];
    foot_world(rot, base, foot, n, e, fw);
    for (int i = 0; i < 12; ++i) st.last_foot[(size_t)i * n + e] = fw[i];
    for (int i = 0; i < 3; ++i) {
        const double b = base[(size_t)i * n + e];
        st.last_base[(size_t)i * n + e] = b;
        for (int r = 0; r < 10; ++r) st.last_base10[(size_t)(3 * r + i) * n + e] = b;     // np.tile(base, (10, 1))
    }
    st.steps[e] = 0;
}

__global__ __launch_bounds__(A1_BLOCK) void a1_reward_step_kernel(RewK k, mg_a1_reward_state st, int n, const double *base,
                                                                  const double *pose, const double *rot, const double *foot,
                                                                  const double *contact, const double *energy, const int32_t *bad,
                                                                  const double *d_yaw_in, double *terms, double *reward,
                                                                  uint8_t *done) {
    const int e = blockIdx.x * A1_BLOCK + threadIdx.x;
    if (e >= n) return;
    const mg_a1_reward_config &c = k.c;
    const int steps = st.steps[e] + 1;                                           // :321
    const double d_yaw = d_yaw_in ? d_yaw_in[e] : 0.0;
    double b[3], v[3];
    for (int i = 0; i < 3; ++i) {
        b[i] = base[(size_t)i * n + e];
        v[i] = (b[i] - st.last_base[(size_t)i * n + e]) / 0.026;                 // :339
    }
    const double roll = pose[e], pitch0 = pose[(size_t)n + e], yaw = pose[2 * (size_t)n + e];
    // torso :475-506
    double vd0, vd1, vd2 = st.vd2[e];
    direction(c, e, d_yaw, b[0], vd0, vd1, vd2);
    st.vd2[e] = vd2;
    double v_ = (v[0] * vd0 + v[1] * vd1) + v[2] * vd2;
    const double v_reward = c.vel_mode == 1 ? exp(-5.0 * fabs(v_ - c.vel_d)) : fmin(c.vel_d, v_);   // :512-518
    const double torso = c.w_torso * re_rot(c, yaw, d_yaw, v_reward);
    const double kk = 1 - c_prec(fmin(v[0], c.vel_d), c.vel_d, c.cw_half);       // :377
    // up :394-409
    int up_f, down_f; double ang;
    env_vec(c, e, b[0], up_f, down_f, ang);
    double pitch = pitch0;
    if (up_f) pitch += fabs(ang);
    else if (down_f) pitch -= fabs(ang);
    const double up = (c.w_up * (1 - c_prec(sqrt(roll * roll + pitch * pitch), 0.0, c.cw_04))) * kk;
    // feet :430-456
    double wd0, wd1, wd2 = st.vd2[(size_t)n + e];
    direction(c, e, d_yaw, b[0], wd0, wd1, wd2);
    st.vd2[(size_t)n + e] = wd2;
    double fw[12];
    foot_world(rot, base, foot, n, e, fw);
    double v_sum = 0.0;
    for (int f = 0; f < 4; ++f) {
        double df[3];
        for (int r = 0; r < 3; ++r) df[r] = (fw[3 * f + r] - st.last_foot[(size_t)(3 * f + r) * n + e]) / 0.026;
        const double vf = (df[0] * wd0 + df[1] * wd1) + df[2] * wd2;
        const double rr = fmin(vf, c.vel_d) / 4.0;
        v_sum += fmin(rr, 1.0 * rr);
    }
    const double feet = c.w_feet * re_rot(c, yaw, d_yaw, v_sum);
    const double tau = (-c.w_tau * energy[e]) * kk;                              // :380
    const double badfoot = -c.w_badfoot * (double)bad[e];                        // :381
    double lose = 0.0;
    for (int f = 0; f < 4; ++f) lose += 1.0 - contact[(size_t)f * n + e];        // :382
    const double footcontact = -c.w_footcontact * fmax(lose - 2, 0.0);
    // terminate :373-381, on last_base10 BEFORE this step's base enters it
    double fz[4], fzsum = 0.0, fzmax = -1e300;
    for (int f = 0; f < 4; ++f) { fz[f] = foot[(size_t)(3 * f + 2) * n + e]; fzsum += fz[f]; fzmax = fmax(fzmax, fz[f]); }
    double base_std = 0.0;
    for (int i = 0; i < 3; ++i) {                                                // np.sum(np.std(last_base10, axis=0))
        double m = 0.0;
        for (int r = 0; r < 10; ++r) m += st.last_base10[(size_t)(3 * r + i) * n + e];
        m /= 10.0;
        double s2 = 0.0;
        for (int r = 0; r < 10; ++r) { const double d = st.last_base10[(size_t)(3 * r + i) * n + e] - m; s2 += d * d; }
        base_std += sqrt(s2 / 10.0);
    }
    const bool is_done = rot[(size_t)8 * n + e] < 0.5 || fzsum / 4.0 > -0.1 || fzmax > 0.0 || (base_std <= 2e-4 && steps >= 10) ||
                         fabs(yaw) > 0.6;
    // rewards: the Param_Dict keys present in info, in dict order (:12,:350-353): torso up feet tau done badfoot footcontact
    double rewards = 0.0;
    rewards += torso; rewards += up; rewards += feet; rewards += tau; rewards += is_done ? -1.0 : 0.0;
    rewards += badfoot; rewards += footcontact;
    reward[e] = c.reward_p * rewards;
    done[e] = (uint8_t)is_done;
    if (terms) {
        terms[e] = torso; terms[(size_t)n + e] = up; terms[2 * (size_t)n + e] = feet; terms[3 * (size_t)n + e] = tau;
        terms[4 * (size_t)n + e] = badfoot; terms[5 * (size_t)n + e] = footcontact;
    }
    // :355-358
    for (int i = 0; i < 3; ++i) {
        st.last_base[(size_t)i * n + e] = b[i];
        for (int r = 9; r >= 1; --r) st.last_base10[(size_t)(3 * r + i) * n + e] = st.last_base10[(size_t)(3 * (r - 1) + i) * n + e];
        st.last_base10[(size_t)i * n + e] = b[i];
    }
    for (int i = 0; i < 12; ++i) st.last_foot[(size_t)i * n + e] = fw[i];
    st.steps[e] = steps;
}

// ---- sensor stack -> observation (robot_sensors.py, locomotion_gym_env.py:621-632) -------------------------------------

__global__ __launch_bounds__(A1_BLOCK) void a1_observation_kernel(mg_a1_sensor_config c, mg_a1_sensor_state st, int n,
                                                                  const double *base, const double *rpy, const double *drpy,
                                                                  const double *angles, const double *contact,
                                                                  const uint8_t *reset_mask, double *obs) {
    const int e = blockIdx.x * A1_BLOCK + threadIdx.x;
    if (e >= n) return;
    double *o = obs + (size_t)e * MG_A1_SENSOR_OBS_DIM;
    const int rm = reset_mask != nullptr ? reset_mask[e] : 0;
    if (rm == 2) return;                                                          // not this robot's call (partial reset): state and row untouched
    const bool was_reset = rm == 1;
    int first = was_reset ? 3 : st.first[e];                                      // IMUSensor.reset / MotorAngleAccSensor.reset
    double cur[3], last[3], yaw_last, yaw_cur;
    for (int i = 0; i < 3; ++i) cur[i] = base[(size_t)i * n + e];
    yaw_cur = rpy[2 * (size_t)n + e];
    if (was_reset) {                                                             // BaseDisplacementSensor.on_reset :298-303
        for (int i = 0; i < 3; ++i) last[i] = cur[i];
        yaw_last = yaw_cur;
    } else {                                                                     // on_step :305-310
        for (int i = 0; i < 3; ++i) last[i] = st.base_cur[(size_t)i * n + e];
        yaw_last = st.yaw[(size_t)n + e];
    }
    for (int i = 0; i < 3; ++i) { st.base_last[(size_t)i * n + e] = last[i]; st.base_cur[(size_t)i * n + e] = cur[i]; }
    st.yaw[e] = yaw_last; st.yaw[(size_t)n + e] = yaw_cur;
    // BaseDisplacementSensor._get_observation :278-296
    const double *nz = st.noise;                                                 // [33][n] draws of sensor_mode["noise"], or none
    double dx = (cur[0] - last[0]) / c.disp_dt, dy = (cur[1] - last[1]) / c.disp_dt, dz = (cur[2] - last[2]) / c.disp_dt;
    if (nz != nullptr) { dx += nz[e]; dy += nz[(size_t)n + e]; dz += nz[2 * (size_t)n + e]; }     // :281-284, before the rotation
    const double cy = cos(yaw_last), sy = sin(yaw_last);
    double d0 = cy * dx + sy * dy, d1 = -sy * dx + cy * dy, d2 = dz;
    if (c.normal) { d0 = (d0 - 0.0) / 0.1; d1 = (d1 - 0.0) / 0.1; d2 = (d2 - 0.0) / 0.1; }
    o[0] = d0; o[1] = d1; o[2] = d2;
    // FootContactSensor :575-576
    for (int f = 0; f < 4; ++f) o[3 + f] = contact[(size_t)f * n + e];
    // IMUSensor._get_observation :387-434
    for (int i = 0; i < 3; ++i) {
        const double r = rpy[(size_t)i * n + e];
        double fr = st.first_rpy[(size_t)i * n + e];
        if (first & 1) { fr = r; st.first_rpy[(size_t)i * n + e] = r; }
        double v = r - fr, w = drpy[(size_t)i * n + e];
        if (nz != nullptr) { v += nz[(size_t)(3 + i) * n + e]; w += nz[(size_t)(6 + i) * n + e]; }   // :399-402
        if (c.normal) { v = (v - 0.0) / 0.1; w = (w - 0.0) / 0.5; }
        o[7 + i] = v; o[10 + i] = w;
    }
    // MotorAngleAccSensor._get_observation :136-158
    const double mean[3] = {0.0, 0.9, -1.8};
    for (int i = 0; i < NM; ++i) {
        double a = angles[(size_t)i * n + e];
        double acc = (first & 2) ? 0.0 : (a - st.last_angle[(size_t)i * n + e]) / c.motor_dt;
        // :146-149: noise after the rate was formed; the NOISY angle is what the next rate starts from
        if (nz != nullptr) { a += nz[(size_t)(9 + i) * n + e]; acc += nz[(size_t)(21 + i) * n + e]; }
        st.last_angle[(size_t)i * n + e] = a;
        double av = a;
        if (c.normal) { av = (a - mean[i % 3]) / 0.1; acc = (acc - 0.0) / 1.0; }
        o[13 + i] = av; o[25 + i] = acc;
    }
    st.first[e] = 0;
}

// ---- ActionFilter (robots/action_filter.py:70-99) ---------------------------------------------------------------------

__global__ __launch_bounds__(A1_BLOCK) void a1_filter_kernel(mg_a1_filter_config c, int n, double *xh, double *yh,
                                                             const double *x, double *y, const uint8_t *init_mask, int mode) {
    const int e = blockIdx.x * A1_BLOCK + threadIdx.x;
    if (e >= n) return;
    const int H = c.hist_len;
    const bool flagged = init_mask == nullptr ? mode == 1 : init_mask[e] != 0;
    if (mode == 1) {                                                             // reset :70-76
        if (flagged)
            for (int k = 0; k < H; ++k)
                for (int j = 0; j < NM; ++j) { xh[((size_t)k * NM + j) * n + e] = 0.0; yh[((size_t)k * NM + j) * n + e] = 0.0; }
        return;
    }
    for (int j = 0; j < NM; ++j) {
        const double xv = x[(size_t)j * n + e];
        double hx[MG_A1_FILTER_MAX_HIST], hy[MG_A1_FILTER_MAX_HIST];
        for (int k = 0; k < H; ++k) {
            hx[k] = flagged ? xv : xh[((size_t)k * NM + j) * n + e];            // init_history :95-99
            hy[k] = flagged ? xv : yh[((size_t)k * NM + j) * n + e];
        }
        // np.sum(..., axis=-1) over H <= 4 contiguous elements is a plain left-to-right sum (numpy pairs only from 8 up)
        double sx = hx[0] * c.b[j][1], sy = hy[0] * c.a[j][1];
        for (int k = 1; k < H; ++k) { sx += hx[k] * c.b[j][k + 1]; sy += hy[k] * c.a[j][k + 1]; }
        const double yv = (xv * c.b[j][0] + sx) - sy;                            // :80-84
        for (int k = H - 1; k >= 1; --k) {                                       // appendleft :85-86
            xh[((size_t)k * NM + j) * n + e] = hx[k - 1];
            yh[((size_t)k * NM + j) * n + e] = hy[k - 1];
        }
        xh[(size_t)j * n + e] = xv;
        yh[(size_t)j * n + e] = yv;
        y[(size_t)j * n + e] = yv;
    }
}

// ---- info entries computed from the control observation (locomotion_gym_env.py:534-545) ------------------------------

__global__ __launch_bounds__(A1_BLOCK) void a1_info_kernel(A1K k, mg_a1_actuator_state st, int n, double *pose, double *rot,
                                                           double *foot, double *angles, double *drpy, double *energy) {
    const int e = blockIdx.x * A1_BLOCK + threadIdx.x;
    if (e >= n) return;
    const mg_a1_actuator_config &c = k.c;
    const double *co = st.control_obs;
    double ang[NM], dot = 0.0;
#pragma unroll
    for (int i = 0; i < NM; ++i) {
        ang[i] = map_to_pi(co[(size_t)i * n + e]);
        dot += co[(size_t)(2 * NM + i) * n + e] * co[(size_t)(NM + i) * n + e];
        if (angles) angles[(size_t)i * n + e] = ang[i];
    }
    if (energy) energy[e] = fabs(dot) * c.time_step * (double)c.action_repeat;
    if (drpy)
        for (int i = 0; i < 3; ++i) drpy[(size_t)i * n + e] = co[(size_t)(3 * NM + 4 + i) * n + e];
    if (pose || rot) {
        const double x = co[(size_t)(3 * NM) * n + e], y = co[(size_t)(3 * NM + 1) * n + e], z = co[(size_t)(3 * NM + 2) * n + e],
                     w = co[(size_t)(3 * NM + 3) * n + e];
        const double roll = atan2(2 * (w * x + y * z), 1 - 2 * (x * x + y * y));
        const double sp = 2 * (w * y - z * x);
        const double pitch = asin(fmax(-1.0, fmin(1.0, sp)));
        const double yaw = atan2(2 * (w * z + x * y), 1 - 2 * (y * y + z * z));
        if (pose) { pose[e] = roll; pose[(size_t)n + e] = pitch; pose[2 * (size_t)n + e] = yaw; }
        if (rot) {      // GetBaseOrientation: getQuaternionFromEuler(rpy), then its matrix
            const double cr = cos(roll / 2), sr = sin(roll / 2), cp = cos(pitch / 2), sq = sin(pitch / 2), cy = cos(yaw / 2), sy = sin(yaw / 2);
            const double qx = sr * cp * cy - cr * sq * sy, qy = cr * sq * cy + sr * cp * sy, qz = cr * cp * sy - sr * sq * cy,
                         qw = cr * cp * cy + sr * sq * sy;
            const double m[9] = {1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy),
                                 2 * (qx * qy + qw * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qw * qx),
                                 2 * (qx * qz - qw * qy), 2 * (qy * qz + qw * qx), 1 - 2 * (qx * qx + qy * qy)};
            for (int i = 0; i < 9; ++i) rot[(size_t)i * n + e] = m[i];
        }
    }
    if (foot) {                                                                  // a1.py:105-123 + HIP_OFFSETS :61-63
        const double com[3] = {-0.012731, -0.002186, -0.000515};
        const double hipx[4] = {0.183, 0.183, -0.183, -0.183}, hipy[4] = {-0.047, 0.047, -0.047, 0.047};
        for (int leg = 0; leg < 4; ++leg) {
            const double ab = ang[3 * leg], hip = ang[3 * leg + 1], knee = ang[3 * leg + 2];
            const double l_up = 0.2, l_low = 0.2, l_hip = 0.08505 * ((leg & 1) ? 1.0 : -1.0);
            const double dist = sqrt(l_up * l_up + l_low * l_low + 2 * l_up * l_low * cos(knee));
            const double swing = hip + knee / 2;
            const double ox = -dist * sin(swing), oz = -dist * cos(swing), oy = l_hip;
            foot[(size_t)(3 * leg) * n + e] = ox + (hipx[leg] + com[0]);
            foot[(size_t)(3 * leg + 1) * n + e] = (cos(ab) * oy - sin(ab) * oz) + (hipy[leg] + com[1]);
            foot[(size_t)(3 * leg + 2) * n + e] = (sin(ab) * oy + cos(ab) * oz) + (0.0 + com[2]);
        }
    }
}

// K logged sub-step observations (mg_walker_params.substep_log) pushed at once; control observation refreshed at the end.
__global__ __launch_bounds__(A1_BLOCK) void a1_receive_log_kernel(A1K k, mg_a1_actuator_state st, int n, const double *log, int K) {
    int e, sub;
    if (!act_lane(n, e, sub)) return;             // four lanes per robot, each pushing the components it owns
    const mg_a1_actuator_config &c = k.c;
    const size_t stride = (size_t)OD * n;
    int count = st.count[e], head = st.head[e];
    for (int s = 0; s < K; ++s) {
        head = count == 0 ? 0 : (head + 1 == c.history_len ? 0 : head + 1);
        if (count < c.history_len) ++count;
        const double *src = log + (size_t)s * stride + e;
        double *slot = st.history + (size_t)head * stride + e;
        double in[ACT_COMPS];
#pragma unroll
        for (int q = 0; q < ACT_COMPS; ++q) {
            const int comp = act_comp(sub, q);
            in[q] = comp >= 0 ? src[(size_t)comp * n] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < ACT_COMPS; ++q) {
            const int comp = act_comp(sub, q);
            if (comp >= 0) slot[(size_t)comp * n] = in[q];
        }
        if (s == K - 1)
#pragma unroll
            for (int m = 0; m < ACT_MOTORS; ++m)      // act_comp(sub, 2 * ACT_MOTORS + m) = the torque of motor 3 sub + m
                st.observed_torque[(size_t)(ACT_MOTORS * sub + m) * n + e] = in[2 * ACT_MOTORS + m];
    }
    if (sub == 0) { st.count[e] = count; st.head[e] = head; }
    const double lat = c.control_latency_env ? c.control_latency_env[e] : c.control_latency;
    const Delay d = delayed(lat, c.time_step, count, head, c.history_len);
    double v[ACT_COMPS];
#pragma unroll
    for (int q = 0; q < ACT_COMPS; ++q) {
        const int comp = act_comp(sub, q);
        v[q] = comp >= 0 ? blend(d, st.history, stride, comp, n, e) : 0.0;
    }
#pragma unroll
    for (int q = 0; q < ACT_COMPS; ++q) {
        const int comp = act_comp(sub, q);
        if (comp >= 0) st.control_obs[(size_t)comp * n + e] = v[q];
    }
}

int check_a1(const mg_a1_actuator_config *cfg, const mg_a1_actuator_state *st, int n) {
    if (!cfg || !st) return mg::set_error(MG_ERR_NULL_POINTER, "a1: NULL descriptor");
    if (n <= 0) return mg::set_error(MG_ERR_BAD_SIZE, "n_envs=%d", n);
    if (!(cfg->time_step > 0) || cfg->action_repeat < 1 || cfg->history_len < 2)
        return mg::set_error(MG_ERR_BAD_CONFIG, "a1: time_step %g action_repeat %d history_len %d", cfg->time_step,
                             cfg->action_repeat, cfg->history_len);
    if (cfg->mode != MG_A1_MODE_POSITION && cfg->mode != MG_A1_MODE_TORQUE && cfg->mode != MG_A1_MODE_HYBRID)
        return mg::set_error(MG_ERR_BAD_CONFIG, "a1: motor control mode %d (PWM is Minitaur-only, laikago_motor.py:120)",
                             cfg->mode);
    if (!st->history || !st->count || !st->head || !st->observed_torque || !st->control_obs)
        return mg::set_error(MG_ERR_NULL_POINTER, "mg_a1_actuator_state has a NULL array");
    return MG_OK;
}


// ---- ObservationWrapper (MonitorEnv.py:77-221): the entries it appends to the sensor observation ---------------------
__global__ __launch_bounds__(A1_BLOCK) void a1_obs_extras_kernel(int n, int flags, int normal, int H, const double *etg_act,
                                                                 const double *etg_obs, const double *pose, const double *d_yaw,
                                                                 double *out, int width) {
    const int i = blockIdx.x * A1_BLOCK + threadIdx.x;
    if (i >= n) return;
    // :89-94
    const double mean[NM] = {2.1505982e-02, 3.6674485e-02, -6.0444288e-02, 2.4625482e-02, 1.5869144e-02, -3.2513142e-02,
                             2.1506395e-02, 3.1869926e-02, -6.0140789e-02, 2.4625063e-02, 1.1628972e-02, -3.2163858e-02};
    const double sd[NM] = {4.5967497e-02, 2.0340437e-01, 3.7410179e-01, 4.6187632e-02, 1.9441207e-01, 3.9488649e-01,
                           4.5966785e-02, 2.0323379e-01, 3.7382501e-01, 4.6188373e-02, 1.9457331e-01, 3.9302582e-01};
    double *o = out + (size_t)i * width;
    if (flags & MG_A1_EXTRA_ETG) {                                                   // :186-190
#pragma unroll
        for (int j = 0; j < NM; ++j) {
            const double a = etg_act[(size_t)j * n + i];
            o[j] = normal ? (a - mean[j]) / sd[j] : a;
        }
        o += NM;
    }
    if (flags & MG_A1_EXTRA_ETG_OBS) {                                               // :192-194
        for (int h = 0; h < H; ++h) o[h] = etg_obs[(size_t)h * n + i];
        o += H;
    }
    if (flags & MG_A1_EXTRA_YAW) {                                                   // :204-211
        const double d = (d_yaw ? d_yaw[i] : 0.0) - pose[(size_t)2 * n + i];
        o[0] = cos(d);
        o[1] = sin(d);
    }
}

}  // namespace

extern "C" int mg_a1_apply_action(const mg_a1_actuator_config *cfg, int32_t n, const mg_a1_actuator_state *st,
                                  const double *command, const double *last_command, double lerp, double *torque,
                                  void *stream) {
    if (int rc = check_a1(cfg, st, n)) return rc;
    MG_REQUIRE_PTR(command);
    MG_REQUIRE_PTR(torque);
    mg::DeviceGuard guard(mg::device_of(st->history));
    A1K k{*cfg};
    hipLaunchKernelGGL(a1_apply_action_kernel, dim3(act_blocks(n)), dim3(A1_BLOCK), 0, (hipStream_t)stream,
                       k, *st, n, command, last_command, lerp, torque);
    return mg::check_launch("a1_apply_action_kernel");
}

extern "C" int mg_a1_receive_observation(const mg_a1_actuator_config *cfg, int32_t n, const mg_a1_actuator_state *st,
                                         const double *q, const double *qd, const double *base_quat,
                                         const double *rpy_rate, const uint8_t *clear_mask, void *stream) {
    if (int rc = check_a1(cfg, st, n)) return rc;
    MG_REQUIRE_PTR(q);
    MG_REQUIRE_PTR(qd);
    MG_REQUIRE_PTR(base_quat);
    MG_REQUIRE_PTR(rpy_rate);
    mg::DeviceGuard guard(mg::device_of(st->history));
    A1K k{*cfg};
    hipLaunchKernelGGL(a1_receive_kernel, dim3(act_blocks(n)), dim3(A1_BLOCK), 0, (hipStream_t)stream, k,
                       *st, n, q, qd, base_quat, rpy_rate, clear_mask);
    return mg::check_launch("a1_receive_kernel");
}

extern "C" int mg_a1_sensors(const mg_a1_actuator_config *cfg, int32_t n, const mg_a1_actuator_state *st,
                             double *motor_angles, double *motor_velocities, double *motor_torques, double *rpy_rate,
                             double *energy, void *stream) {
    if (int rc = check_a1(cfg, st, n)) return rc;
    mg::DeviceGuard guard(mg::device_of(st->history));
    A1K k{*cfg};
    hipLaunchKernelGGL(a1_sensors_kernel, dim3((n + A1_BLOCK - 1) / A1_BLOCK), dim3(A1_BLOCK), 0, (hipStream_t)stream, k,
                       *st, n, motor_angles, motor_velocities, motor_torques, rpy_rate, energy);
    return mg::check_launch("a1_sensors_kernel");
}

extern "C" int mg_a1_etg_action(const mg_a1_etg_config *cfg, int32_t n, double *last_etg_act, const double *action,
                                const double *t, double *command, double *etg_obs, void *stream) {
    if (!cfg) return mg::set_error(MG_ERR_NULL_POINTER, "mg_a1_etg_action: cfg is NULL");
    if (n <= 0) return mg::set_error(MG_ERR_BAD_SIZE, "n_envs=%d", n);
    if (cfg->enabled && (cfg->H < 1 || cfg->H > MG_A1_ETG_MAX_H || !(cfg->sigma_sq > 0)))
        return mg::set_error(MG_ERR_BAD_CONFIG, "ETG: H %d (max %d) sigma_sq %g", cfg->H, MG_A1_ETG_MAX_H, cfg->sigma_sq);
    if (cfg->action_space < 0 || cfg->action_space > 3)
        return mg::set_error(MG_ERR_BAD_CONFIG, "LaikagoPoseOffsetGenerator action_space %d", cfg->action_space);
    if (cfg->enabled) MG_REQUIRE_PTR(last_etg_act);
    if (action != nullptr) MG_REQUIRE_PTR(command);
    mg::DeviceGuard guard(mg::device_of(action ? (const void *)command : (const void *)last_etg_act));
    EtgK k{*cfg};
    hipLaunchKernelGGL(a1_etg_kernel, dim3((n + A1_BLOCK - 1) / A1_BLOCK), dim3(A1_BLOCK), 0, (hipStream_t)stream, k, n,
                       last_etg_act, action, t, command, etg_obs);
    return mg::check_launch("a1_etg_kernel");
}

namespace {
int check_reward(const mg_a1_reward_config *cfg, const mg_a1_reward_state *st, int n) {
    if (!cfg || !st) return mg::set_error(MG_ERR_NULL_POINTER, "a1 reward: NULL descriptor");
    if (n <= 0) return mg::set_error(MG_ERR_BAD_SIZE, "n_envs=%d", n);
    if (cfg->n_segments < 0 || cfg->n_segments > MG_A1_MAX_SEGMENTS)
        return mg::set_error(MG_ERR_BAD_SIZE, "a1 reward: %d terrain segments (max %d)", cfg->n_segments, MG_A1_MAX_SEGMENTS);
    if (!st->last_base || !st->last_base10 || !st->last_foot || !st->vd2 || !st->steps)
        return mg::set_error(MG_ERR_NULL_POINTER, "mg_a1_reward_state has a NULL array");
    return MG_OK;
}
}  // namespace

extern "C" int mg_a1_reward_reset(const mg_a1_reward_config *cfg, int32_t n, const mg_a1_reward_state *st, const double *base,
                                  const double *rot_mat, const double *footposition, const uint8_t *mask, void *stream) {
    if (int rc = check_reward(cfg, st, n)) return rc;
    MG_REQUIRE_PTR(base);
    MG_REQUIRE_PTR(rot_mat);
    MG_REQUIRE_PTR(footposition);
    mg::DeviceGuard guard(mg::device_of(st->last_base));
    hipLaunchKernelGGL(a1_reward_reset_kernel, dim3((n + A1_BLOCK - 1) / A1_BLOCK), dim3(A1_BLOCK), 0, (hipStream_t)stream,
                       *st, n, base, rot_mat, footposition, mask);
    return mg::check_launch("a1_reward_reset_kernel");
}

extern "C" int mg_a1_reward_step(const mg_a1_reward_config *cfg, int32_t n, const mg_a1_reward_state *st, const double *base,
                                 const double *pose, const double *rot_mat, const double *footposition,
                                 const double *real_contact, const double *energy, const int32_t *bad_contacts,
                                 const double *d_yaw, double *terms, double *reward, uint8_t *done, void *stream) {
    if (int rc = check_reward(cfg, st, n)) return rc;
    MG_REQUIRE_PTR(base);
    MG_REQUIRE_PTR(pose);
    MG_REQUIRE_PTR(rot_mat);
    MG_REQUIRE_PTR(footposition);
    MG_REQUIRE_PTR(real_contact);
    MG_REQUIRE_PTR(energy);
    MG_REQUIRE_PTR(bad_contacts);
    MG_REQUIRE_PTR(reward);
    MG_REQUIRE_PTR(done);
    mg::DeviceGuard guard(mg::device_of(st->last_base));
    RewK k{*cfg};
    hipLaunchKernelGGL(a1_reward_step_kernel, dim3((n + A1_BLOCK - 1) / A1_BLOCK), dim3(A1_BLOCK), 0, (hipStream_t)stream, k,
                       *st, n, base, pose, rot_mat, footposition, real_contact, energy, bad_contacts, d_yaw, terms, reward, done);
    return mg::check_launch("a1_reward_step_kernel");
}

extern "C" int mg_a1_observation(const mg_a1_sensor_config *cfg, int32_t n, const mg_a1_sensor_state *st, const double *base,
                                 const double *rpy, const double *drpy, const double *motor_angles, const double *contact,
                                 const uint8_t *reset_mask, double *obs, void *stream) {
    if (!cfg || !st) return mg::set_error(MG_ERR_NULL_POINTER, "mg_a1_observation: NULL descriptor");
    if (n <= 0) return mg::set_error(MG_ERR_BAD_SIZE, "n_envs=%d", n);
    if (!(cfg->disp_dt > 0) || !(cfg->motor_dt > 0)) return mg::set_error(MG_ERR_BAD_CONFIG, "a1 sensors: dt");
    if (!st->base_last || !st->base_cur || !st->yaw || !st->first_rpy || !st->last_angle || !st->first)
        return mg::set_error(MG_ERR_NULL_POINTER, "mg_a1_sensor_state has a NULL array");
    MG_REQUIRE_PTR(base);
    MG_REQUIRE_PTR(rpy);
    MG_REQUIRE_PTR(drpy);
    MG_REQUIRE_PTR(motor_angles);
    MG_REQUIRE_PTR(contact);
    MG_REQUIRE_PTR(obs);
    mg::DeviceGuard guard(mg::device_of(st->base_cur));
    hipLaunchKernelGGL(a1_observation_kernel, dim3((n + A1_BLOCK - 1) / A1_BLOCK), dim3(A1_BLOCK), 0, (hipStream_t)stream, *cfg,
                       *st, n, base, rpy, drpy, motor_angles, contact, reset_mask, obs);
    return mg::check_launch("a1_observation_kernel");
}

extern "C" int mg_a1_observation_extras(int32_t n, int32_t flags, int32_t normal, int32_t etg_h, const double *etg_act,
                                        const double *etg_obs, const double *pose, const double *d_yaw, double *out,
                                        void *stream) {
    if (n <= 0) return mg::set_error(MG_ERR_BAD_SIZE, "n_envs=%d", n);
    if (flags <= 0 || (flags & ~(MG_A1_EXTRA_ETG | MG_A1_EXTRA_ETG_OBS | MG_A1_EXTRA_YAW)))
        return mg::set_error(MG_ERR_BAD_CONFIG, "mg_a1_observation_extras: flags=0x%x", flags);
    if ((flags & MG_A1_EXTRA_ETG_OBS) && (etg_h < 1 || etg_h > MG_A1_ETG_MAX_H))
        return mg::set_error(MG_ERR_BAD_SIZE, "mg_a1_observation_extras: etg_h=%d (1..%d)", etg_h, MG_A1_ETG_MAX_H);
    if ((flags & MG_A1_EXTRA_ETG) && !etg_act) return mg::set_error(MG_ERR_NULL_POINTER, "mg_a1_observation_extras: etg_act is NULL");
    if ((flags & MG_A1_EXTRA_ETG_OBS) && !etg_obs) return mg::set_error(MG_ERR_NULL_POINTER, "mg_a1_observation_extras: etg_obs is NULL");
    if ((flags & MG_A1_EXTRA_YAW) && !pose) return mg::set_error(MG_ERR_NULL_POINTER, "mg_a1_observation_extras: pose is NULL");
    MG_REQUIRE_PTR(out);
    const int width = ((flags & MG_A1_EXTRA_ETG) ? 12 : 0) + ((flags & MG_A1_EXTRA_ETG_OBS) ? etg_h : 0) + ((flags & MG_A1_EXTRA_YAW) ? 2 : 0);
    mg::DeviceGuard guard(mg::device_of(out));
    hipLaunchKernelGGL(a1_obs_extras_kernel, dim3((n + A1_BLOCK - 1) / A1_BLOCK), dim3(A1_BLOCK), 0, (hipStream_t)stream, n, flags,
                       normal, etg_h, etg_act, etg_obs, pose, d_yaw, out, width);
    return mg::check_launch("a1_obs_extras_kernel");
}

extern "C" int mg_a1_action_filter(const mg_a1_filter_config *cfg, int32_t n, double *xhist, double *yhist, const double *x,
                                   double *y, const uint8_t *init_mask, int32_t mode, void *stream) {
    if (!cfg) return mg::set_error(MG_ERR_NULL_POINTER, "mg_a1_action_filter: cfg is NULL");
    if (n <= 0) return mg::set_error(MG_ERR_BAD_SIZE, "n_envs=%d", n);
    if (cfg->hist_len < 1 || cfg->hist_len > MG_A1_FILTER_MAX_HIST)
        return mg::set_error(MG_ERR_BAD_CONFIG, "action filter history %d (1..%d)", cfg->hist_len, MG_A1_FILTER_MAX_HIST);
    MG_REQUIRE_PTR(xhist);
    MG_REQUIRE_PTR(yhist);
    if (mode == 0) { MG_REQUIRE_PTR(x); MG_REQUIRE_PTR(y); }
    mg::DeviceGuard guard(mg::device_of(xhist));
    hipLaunchKernelGGL(a1_filter_kernel, dim3((n + A1_BLOCK - 1) / A1_BLOCK), dim3(A1_BLOCK), 0, (hipStream_t)stream, *cfg, n,
                       xhist, yhist, x, y, init_mask, mode);
    return mg::check_launch("a1_filter_kernel");
}

extern "C" int mg_a1_info(const mg_a1_actuator_config *cfg, int32_t n, const mg_a1_actuator_state *st, double *pose,
                          double *rot_mat, double *footposition, double *joint_angle, double *drpy, double *energy,
                          void *stream) {
    if (int rc = check_a1(cfg, st, n)) return rc;
    mg::DeviceGuard guard(mg::device_of(st->history));
    A1K k{*cfg};
    hipLaunchKernelGGL(a1_info_kernel, dim3((n + A1_BLOCK - 1) / A1_BLOCK), dim3(A1_BLOCK), 0, (hipStream_t)stream, k, *st, n,
                       pose, rot_mat, footposition, joint_angle, drpy, energy);
    return mg::check_launch("a1_info_kernel");
}

extern "C" int mg_a1_receive_and_apply(const mg_a1_actuator_config *cfg, int32_t n, const mg_a1_actuator_state *st,
                                       const double *q, const double *qd, const double *base_quat, const double *rpy_rate,
                                       const double *command, const double *last_command, double lerp, double *torque,
                                       void *stream) {
    if (int rc = check_a1(cfg, st, n)) return rc;
    MG_REQUIRE_PTR(q);
    MG_REQUIRE_PTR(qd);
    MG_REQUIRE_PTR(base_quat);
    MG_REQUIRE_PTR(rpy_rate);
    MG_REQUIRE_PTR(command);
    MG_REQUIRE_PTR(torque);
    mg::DeviceGuard guard(mg::device_of(st->history));
    A1K k{*cfg};
    hipLaunchKernelGGL(a1_receive_apply_kernel, dim3(act_blocks(n)), dim3(A1_BLOCK), 0, (hipStream_t)stream, k,
                       *st, n, q, qd, base_quat, rpy_rate, command, last_command, lerp, torque);
    return mg::check_launch("a1_receive_apply_kernel");
}

extern "C" int mg_a1_receive_log(const mg_a1_actuator_config *cfg, int32_t n, const mg_a1_actuator_state *st, const double *log,
                                 int32_t n_substeps, void *stream) {
    if (int rc = check_a1(cfg, st, n)) return rc;
    MG_REQUIRE_PTR(log);
    if (n_substeps < 1) return mg::set_error(MG_ERR_BAD_SIZE, "n_substeps=%d", n_substeps);
    mg::DeviceGuard guard(mg::device_of(st->history));
    A1K k{*cfg};
    hipLaunchKernelGGL(a1_receive_log_kernel, dim3(act_blocks(n)), dim3(A1_BLOCK), 0, (hipStream_t)stream, k, *st,
                       n, log, n_substeps);
    return mg::check_launch("a1_receive_log_kernel");
}
