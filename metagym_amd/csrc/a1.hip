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
// Mapping: one lane per robot, SoA arrays ([component][N]) so a wave's loads and stores are contiguous. The path is
// HBM-bound: per sub-step a robot writes one 43-double observation into its history ring and reads two ring entries
// for the control observation (+ two 24-double halves when the PD latency is not zero) — about 1.4 KB.
#include "mg_common.h"

namespace {

constexpr int A1_BLOCK = 256;
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

__global__ __launch_bounds__(A1_BLOCK) void a1_apply_action_kernel(A1K k, mg_a1_actuator_state st, int n,
                                                                   const double *command, const double *last_command,
                                                                   double lerp, double *torque) {
    const int e = blockIdx.x * A1_BLOCK + threadIdx.x;
    if (e >= n) return;
    const mg_a1_actuator_config &c = k.c;
    const size_t stride = (size_t)OD * n;
    const double pd_lat = c.pd_latency_env ? c.pd_latency_env[e] : c.pd_latency;
    const Delay d = delayed(pd_lat, c.time_step, st.count[e], st.head[e], c.history_len);
    const int cdim = c.mode == MG_A1_MODE_HYBRID ? 5 * NM : NM;
    auto cmd = [&](int i) {      // ProcessAction: last + lerp * (action - last)
        const double a = command[(size_t)i * n + e];
        if (last_command == nullptr) return a;
        const double l = last_command[(size_t)i * n + e];
        return l + lerp * (a - l);
    };
    (void)cdim;
#pragma unroll
    for (int i = 0; i < NM; ++i) {
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

__global__ __launch_bounds__(A1_BLOCK) void a1_receive_kernel(A1K k, mg_a1_actuator_state st, int n, const double *q,
                                                              const double *qd, const double *quat, const double *rate,
                                                              const uint8_t *clear_mask) {
    const int e = blockIdx.x * A1_BLOCK + threadIdx.x;
    if (e >= n) return;
    const mg_a1_actuator_config &c = k.c;
    const size_t stride = (size_t)OD * n;
    int count = st.count[e], head = st.head[e];
    if (clear_mask != nullptr && clear_mask[e]) count = 0;                          // _observation_history.clear()
    head = count == 0 ? 0 : (head + 1 == c.history_len ? 0 : head + 1);             // appendleft
    if (count < c.history_len) ++count;
    double *slot = st.history + (size_t)head * stride + e;
#pragma unroll
    for (int i = 0; i < NM; ++i) {
        // GetTrueMotorAngles (:743-753): (angle - offset) * direction with offset 0, direction 1 (a1.py:43-49)
        slot[(size_t)i * n] = (q[(size_t)i * n + e] - 0.0) * 1.0;
        slot[(size_t)(NM + i) * n] = qd[(size_t)i * n + e] * 1.0;
        slot[(size_t)(2 * NM + i) * n] = st.observed_torque[(size_t)i * n + e];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) slot[(size_t)(3 * NM + i) * n] = quat[(size_t)i * n + e];
#pragma unroll
    for (int i = 0; i < 3; ++i) slot[(size_t)(3 * NM + 4 + i) * n] = rate[(size_t)i * n + e];
    st.count[e] = count;
    st.head[e] = head;
    // a lane reads back only what it wrote itself (its own column of the ring): no fence needed
    const double lat = c.control_latency_env ? c.control_latency_env[e] : c.control_latency;
    const Delay d = delayed(lat, c.time_step, count, head, c.history_len);
    for (int comp = 0; comp < OD; ++comp) st.control_obs[(size_t)comp * n + e] = blend(d, st.history, stride, comp, n, e);
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

}  // namespace

extern "C" int mg_a1_apply_action(const mg_a1_actuator_config *cfg, int32_t n, const mg_a1_actuator_state *st,
                                  const double *command, const double *last_command, double lerp, double *torque,
                                  void *stream) {
    if (int rc = check_a1(cfg, st, n)) return rc;
    MG_REQUIRE_PTR(command);
    MG_REQUIRE_PTR(torque);
    mg::DeviceGuard guard(mg::device_of(st->history));
    A1K k{*cfg};
    hipLaunchKernelGGL(a1_apply_action_kernel, dim3((n + A1_BLOCK - 1) / A1_BLOCK), dim3(A1_BLOCK), 0, (hipStream_t)stream,
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
    hipLaunchKernelGGL(a1_receive_kernel, dim3((n + A1_BLOCK - 1) / A1_BLOCK), dim3(A1_BLOCK), 0, (hipStream_t)stream, k,
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
