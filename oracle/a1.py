"""CPU restatement of the Quadrupedal (A1) actuation path — TEST INFRASTRUCTURE (the checker; the product never
imports it). Vectorised over robots in numpy float64, one function per reference method, each citing the lines of
/root/reference/metagym/quadrupedal it follows. Pinned: tests/test_oracle_a1.py compares it bit for bit with
tests/golden/a1_actuation.npz, recorded from the unmodified reference by oracle/gen_golden_a1.py."""
import math

import numpy as np

NUM_MOTORS, OBS_DIM = 12, 43
POSITION, TORQUE, HYBRID = 1, 2, 3          # robots/robot_config.py:13-27
TWO_PI = 2 * math.pi                        # robots/minitaur.py:52


def map_to_minus_pi_to_pi(angles):
    """MapToMinusPiToPi, robots/minitaur.py:56-73."""
    m = np.fmod(angles, TWO_PI)
    m = np.where(m >= math.pi, m - TWO_PI, np.where(m < -math.pi, m + TWO_PI, m))
    return m


class A1Actuation(object):
    """N robots. history[k] is the k-th newest true observation (deque.appendleft order, robots/minitaur.py:1201)."""

    def __init__(self, n, time_step=0.002, action_repeat=13, control_latency=0.002, pd_latency=0.0, mode=POSITION,
                 kp=(80.0, 80.0, 80.0) * 4, kd=(1.0, 2.0, 2.0) * 4, strength=1.0, torque_limit=33.5, interpolate=False,
                 clip=False, history_len=100):
        self.n, self.dt, self.repeat, self.mode = n, float(time_step), int(action_repeat), mode
        self.control_latency = np.broadcast_to(np.asarray(control_latency, np.float64), (n,)).copy()
        self.pd_latency = np.broadcast_to(np.asarray(pd_latency, np.float64), (n,)).copy()
        self.kp = np.broadcast_to(np.asarray(kp, np.float64), (n, NUM_MOTORS)).copy()
        self.kd = np.broadcast_to(np.asarray(kd, np.float64), (n, NUM_MOTORS)).copy()
        self.strength = np.broadcast_to(np.asarray(strength, np.float64), (NUM_MOTORS,)).copy()
        self.torque_limit = None if torque_limit is None else np.broadcast_to(np.asarray(torque_limit, np.float64), (NUM_MOTORS,)).copy()
        self.interpolate, self.clip, self.hist_len = interpolate, clip, history_len
        self.history = np.zeros((history_len, n, OBS_DIM))
        self.count = np.zeros(n, np.int64)
        self.observed_torque = np.zeros((n, NUM_MOTORS))          # robots/minitaur.py:133
        self.control_obs = np.zeros((n, OBS_DIM))
        self.last_action = None

    def reset(self, mask=None):
        """Minitaur.Reset robots/minitaur.py:434-441."""
        m = np.ones(self.n, bool) if mask is None else np.asarray(mask, bool)
        self.count[m] = 0
        self.observed_torque[m] = 0.0
        self.last_action = None

    def delayed(self, latency):
        """_GetDelayedObservation robots/minitaur.py:1205-1226, per robot."""
        out = np.empty((self.n, OBS_DIM))
        for e in range(self.n):
            lat, cnt, h = latency[e], int(self.count[e]), self.history[:, e]
            if lat <= 0 or cnt == 1:
                out[e] = h[0]
                continue
            n_steps_ago = int(lat / self.dt)
            if n_steps_ago + 1 >= cnt:
                out[e] = h[cnt - 1]
                continue
            remaining_latency = lat - n_steps_ago * self.dt
            blend_alpha = remaining_latency / self.dt
            out[e] = (1.0 - blend_alpha) * h[n_steps_ago] + blend_alpha * h[n_steps_ago + 1]
        return out

    def process_action(self, action, substep):
        """ProcessAction robots/minitaur.py:1419-1436."""
        if self.interpolate and self.last_action is not None:
            lerp = float(substep + 1) / self.repeat
            return self.last_action + lerp * (action - self.last_action)
        return action

    def apply_action(self, commands):
        """A1.ApplyAction robots/a1.py:451-483 -> Minitaur.ApplyAction robots/minitaur.py:906-955 ->
        LaikagoMotorModel.convert_to_torque robots/laikago_motor.py:92-169. Returns the applied torques [n, 12]."""
        commands = np.asarray(commands, np.float64)
        if self.clip:                                                       # robots/a1.py:465-483
            cur = map_to_minus_pi_to_pi(self.control_obs[:, :NUM_MOTORS])
            commands = np.clip(commands, cur - 0.2, cur + 0.2)
        pd = self.delayed(self.pd_latency)                                  # _GetPDObservation :1228-1232
        q, qdot = pd[:, :NUM_MOTORS], pd[:, NUM_MOTORS:2 * NUM_MOTORS]
        if self.mode == TORQUE:                                             # laikago_motor.py:124-128
            t = self.strength * commands
        else:
            if self.mode == POSITION:                                       # :136-141
                kp, kd, q_des = self.kp, self.kd, commands
                qd_des, extra = np.full(NUM_MOTORS, 0), np.full(NUM_MOTORS, 0)
            else:                                                           # HYBRID :142-153
                kp, kd = commands[:, 1::5], commands[:, 3::5]
                q_des, qd_des, extra = commands[:, 0::5], commands[:, 2::5], commands[:, 4::5]
            t = -1 * (kp * (q - q_des)) - kd * (qdot - qd_des) + extra      # :157-158
            t = self.strength * t                                           # :159
            if self.torque_limit is not None:                               # :163-168
                t = np.clip(t, -1.0 * self.torque_limit, self.torque_limit)
        self.observed_torque = t.copy()                                     # robots/minitaur.py:930
        return np.multiply(t, np.ones(NUM_MOTORS))                          # motor_direction, robots/a1.py:43

    def receive_observation(self, q, qd, quat, rpy_rate, clear_mask=None):
        """ReceiveObservation robots/minitaur.py:1184-1203 with GetTrueObservation :1175-1182."""
        if clear_mask is not None:
            self.count[np.asarray(clear_mask, bool)] = 0
        obs = np.concatenate([np.multiply(np.asarray(q) - np.zeros(NUM_MOTORS), np.ones(NUM_MOTORS)),
                              np.multiply(qd, np.ones(NUM_MOTORS)), self.observed_torque, quat, rpy_rate], axis=1)
        fresh = self.count == 0
        self.history[1:] = self.history[:-1].copy()                         # appendleft on a deque(maxlen=100)
        self.history[0] = obs
        self.count = np.minimum(np.where(fresh, 1, self.count + 1), self.hist_len)
        self.control_obs = self.delayed(self.control_latency)               # _GetControlObservation :1234-1237

    def sensors(self):
        """GetMotorAngles / Velocities / Torques robots/minitaur.py:755-810, GetBaseRollPitchYawRate :874-885,
        GetEnergyConsumptionPerControlStep :812-820 (noise stdev 0, :48)."""
        c = self.control_obs
        ang = map_to_minus_pi_to_pi(c[:, :NUM_MOTORS])
        vel, tor = c[:, NUM_MOTORS:2 * NUM_MOTORS], c[:, 2 * NUM_MOTORS:3 * NUM_MOTORS]
        energy = np.array([np.abs(np.dot(tor[e], vel[e])) * self.dt * self.repeat for e in range(self.n)])
        return ang, vel, tor, c[:, 3 * NUM_MOTORS + 4:3 * NUM_MOTORS + 7], energy


def from_golden(g, name, n=1):
    """An A1Actuation configured like golden case `name` (tests/golden/a1_actuation.npz)."""
    dt, repeat, clat, plat, interp, clip, mode, _ = g[name + "/config"]
    return A1Actuation(n, dt, int(repeat), clat, plat, int(mode), g[name + "/kp"], g[name + "/kd"], g[name + "/strength"],
                       g[name + "/torque_limit"], bool(interp), bool(clip))


# =====================================================================================================================
# Control-side wrappers of A1GymEnv.step: the ETG action path and the reward shaping (pinned by
# tests/golden/a1_control.npz, oracle/gen_golden_a1_control.py)
# =====================================================================================================================

BASE_FOOT = np.array([0.18, -0.15, -0.23, 0.18, 0.148, -0.23, -0.18, -0.14, -0.23, -0.18, 0.135, -0.23])   # ETG_model.py:5-6
COM_OFFSET = -np.array([0.012731, 0.002186, 0.000515])                                                     # a1.py:60
HIP_OFFSETS = np.array([[0.183, -0.047, 0.], [0.183, 0.047, 0.], [-0.183, -0.047, 0.], [-0.183, 0.047, 0.]]) + COM_OFFSET
POSE_ORI = np.array([0, 0.9, -1.8] * 4)                                                                    # ETG_model.py:83


def leg_ik(foot, side):
    """Leg inverse kinematics of robots/a1.py:88-102 (foot position in the hip frame -> abduction, hip, knee angles), same
    operations in the same order: squares as `**2`, knee from the law of cosines, the virtual leg length from the knee angle,
    hip from the fore-aft offset, abduction from atan2 of the lateral / vertical pair. `side` = +1 left, -1 right."""
    upper = lower = 0.2
    hip = 0.08505 * side
    fx, fy, fz = foot[0], foot[1], foot[2]
    with np.errstate(invalid="ignore"):
        knee = -np.arccos((fx**2 + fy**2 + fz**2 - hip**2 - lower**2 - upper**2) / (2 * lower * upper))
        leg = np.sqrt(upper**2 + lower**2 + 2 * upper * lower * np.cos(knee))
        hip_angle = np.arcsin(-fx / leg) - knee / 2
    swing = np.cos(hip_angle + knee / 2)
    cos_term = hip * fy - leg * swing * fz
    sin_term = leg * swing * fy + hip * fz
    return np.array([np.arctan2(sin_term, cos_term), hip_angle, knee])


class EtgActionPath(object):
    """One robot's ETGWrapper (MonitorEnv.py:222-273) over TrajectoryGeneratorWrapperEnv + LaikagoPoseOffsetGenerator."""

    def __init__(self, w, b, enabled=True, T=0.5, T2_ratio=0.5, H=20, sigma_sq=0.04, amp=0.2, pose_mode=False, gallop=False,
                 action_space=0, etg_weight=1):
        self.enabled, self.T, self.T2, self.H, self.sigma_sq, self.amp = enabled, T, T2_ratio, H, sigma_sq, amp
        self.phase = np.array([-np.pi / 2, 0])                        # MonitorEnv.py:236
        self.omega = 2.0 * np.pi / T                                   # ETG_model.py:20
        self.u = np.asarray([self.forward(h * T / (H - 0.9)) for h in range(H)]).reshape(-1, 2)      # :22-25
        self.w, self.b, self.pose_mode, self.gallop = np.asarray(w), np.asarray(b), pose_mode, gallop
        self.action_space, self.weight = action_space, etg_weight
        self.pose = np.array([0, 0.9, -1.8] * 4, dtype=np.float64)     # laikago_pose_utils.py:17-19
        self.last_etg_act = np.zeros(12)
        self.retries = 0

    def forward(self, t):                                             # ETG_layer.forward :28-32
        return np.asarray([self.amp * np.sin(self.phase[i] + t * self.omega) for i in range(2)]).reshape(-1)

    def rbf(self, x):                                                 # ETG_layer.update2 :48-51
        return np.asarray([np.exp(-(np.sum(np.power(x - self.u[i], 2)) / self.sigma_sq)) for i in range(self.H)]).reshape(-1)

    def etg(self, t):
        """update2 + ETG_model.forward + act_clip: returns (state[0], clipped reference action * weight)."""
        r, r2 = self.rbf(self.forward(t)), self.rbf(self.forward(t + self.T2 * self.T))
        act1 = self.w.dot(r.reshape(-1, 1)).reshape(-1) + self.b       # ETG_model.py:99-103
        act2 = self.w.dot(r2.reshape(-1, 1)).reshape(-1) + self.b
        new_act = np.zeros(12)
        if self.gallop:
            new_act[:3], new_act[3:6], new_act[6:9], new_act[9:] = act1, act1, act2, act2
        else:
            new_act[:3], new_act[3:6], new_act[6:9], new_act[9:] = act1, act2, act2, act1
        if self.pose_mode:                                            # act_clip :118-120
            act = np.tanh(new_act) * np.array([0.1, 0.7, 0.7] * 4)
        else:                                                         # :121-130
            act = np.zeros(12)
            for i in range(4):
                delta = new_act[i * 3:(i + 1) * 3].copy()
                while True:
                    angle = leg_ik(delta + BASE_FOOT[i * 3:(i + 1) * 3] - HIP_OFFSETS[i], (-1) ** (i + 1))   # a1.py:509-511
                    angle = np.multiply(angle - np.zeros(3), np.ones(3))                               # :514-517
                    if np.sum(np.isnan(angle)) == 0:
                        break
                    delta *= 0.95
                    self.retries += 1
                act[i * 3:(i + 1) * 3] = angle
            act -= POSE_ORI
        return r, act * self.weight

    def reset(self, t=0.0):
        """ETGWrapper.reset :246-259."""
        if not self.enabled:
            return None
        obs, self.last_etg_act = self.etg(t)
        return obs

    def generator(self, a):
        """LaikagoPoseOffsetGenerator.get_action simple_openloop.py:144-165."""
        if self.action_space <= 1:
            return self.pose + a
        new_action = np.zeros(12)
        new_action[6:9] = a[3:6]
        new_action[9:12] = a[:3]
        new_action += a
        return self.pose + new_action

    def step(self, action, t):
        """ETGWrapper.step :261-273 -> TrajectoryGeneratorWrapperEnv.step: returns (motor command, ETG_obs or None)."""
        if not self.enabled:
            return self.generator(np.asarray(action)), None
        total = np.asarray(action).reshape(-1) + self.last_etg_act
        obs, self.last_etg_act = self.etg(t)
        return self.generator(total), obs


class RewardShaping(object):
    """One robot's RewardShaping wrapper, MonitorEnv.py:275-519 (vel_mode "max" or "equal", :512-518)."""

    def __init__(self, param, reward_p=1.0, vel_d=0.6, segments=((-100, 100, 1, 0, 0.0),), vel_mode="max"):
        assert vel_mode in ("max", "equal")
        self.vel_mode = vel_mode
        self.p = dict(zip(("torso", "up", "feet", "tau", "badfoot", "footcontact"), param))
        self.reward_p, self.vel_d, self.segments = reward_p, vel_d, [tuple(s) for s in segments]
        self.vd_torso, self.vd_feet = [1, 0, 0], [1, 0, 0]              # the mutable default arguments of :475 and :430
        self.steps = 0

    def env_vec(self, posex):                                          # :328-333 (and four more copies)
        for x0, x1, up, down, ang in self.segments:
            if posex + 0.2 >= x0 and posex + 0.2 <= x1:
                return up, down, ang
        return 0, 0, 0.0

    @staticmethod
    def c_prec(v, t, m):                                               # :421-425
        w = np.arctanh(np.sqrt(0.95)) / m
        return np.tanh(np.power((v - t) * w, 2))

    @staticmethod
    def foot_world(base, rot_mat, foot):                               # get_foot_world :458-473
        f = np.array(foot).transpose()
        return (np.array(rot_mat).reshape(-1, 3).dot(f) + np.array(base).reshape(-1, 1)).transpose()

    def reset(self, base, rot_mat, foot):                              # :305-318
        self.steps = 0
        self.last_basepose = np.array(base)
        self.last_foot = self.foot_world(base, rot_mat, foot)
        self.last_base10 = np.tile(base, (10, 1))

    def direction(self, vd, d_yaw, base):                              # the block shared by re_torso :481-497 and re_feet :431-446
        vd[0], vd[1] = np.cos(d_yaw), np.sin(d_yaw)
        up, down, ang = self.env_vec(base[0])
        if up:
            vd[0] *= abs(np.cos(ang)); vd[1] *= abs(np.cos(ang)); vd[2] = abs(np.sin(ang))
        elif down:
            vd[0] *= abs(np.cos(ang)); vd[1] *= abs(np.cos(ang)); vd[2] = -abs(np.sin(ang))
        return vd

    def re_rot(self, yaw, d_yaw, r):                                   # :411-419
        k = max(1 - self.c_prec(yaw, d_yaw, 0.5), 1 - self.c_prec(yaw, d_yaw + 2 * np.pi, 0.5),
                1 - self.c_prec(yaw, d_yaw - 2 * np.pi, 0.5))
        return min(k * r, r)

    def step(self, base, pose, rot_mat, foot, contact, energy, bad, d_yaw=0):
        """RewardShaping.step :320-366. Returns (terms[6], reward, done)."""
        self.steps += 1
        base, pose = np.array(base), np.array(pose)
        v = (base - self.last_basepose) / 0.026
        # torso :475-506
        vd = self.direction(self.vd_torso, d_yaw, base)
        v_ = v[0] * vd[0] + v[1] * vd[1] + v[2] * vd[2]
        v_reward = min(self.vel_d, v_) if self.vel_mode == "max" else np.exp(-5 * abs(v_ - self.vel_d))      # :512-518
        torso = self.p["torso"] * self.re_rot(pose[-1], d_yaw, v_reward)
        k = 1 - self.c_prec(min(v[0], self.vel_d), self.vel_d, 0.5)
        # up :394-409
        up_flag, down_flag, ang = self.env_vec(base[0])
        roll, pitch = pose[0], pose[1]
        if up_flag:
            pitch += abs(ang)
        elif down_flag:
            pitch -= abs(ang)
        up = self.p["up"] * (1 - self.c_prec(np.sqrt(roll ** 2 + pitch ** 2), 0, 0.4)) * k
        # feet :430-456
        vd = self.direction(self.vd_feet, d_yaw, base)
        fw = self.foot_world(base, rot_mat, foot)
        d_foot = (fw - self.last_foot) / 0.026
        v_sum = 0
        for i in range(4):
            vf = d_foot[i]
            v_ = vf[0] * vd[0] + vf[1] * vd[1] + vf[2] * vd[2]
            r = min(v_, self.vel_d) / 4.0
            v_sum += min(r, 1.0 * r)
        feet = self.p["feet"] * self.re_rot(pose[-1], d_yaw, v_sum)
        tau = -self.p["tau"] * energy * k
        badfoot = -self.p["badfoot"] * bad
        lose = np.sum(1.0 - np.array(contact))
        footcontact = -self.p["footcontact"] * max(lose - 2, 0)
        # terminate :373-381 (last_base10 not yet updated)
        footz = np.array(foot)[:, -1]
        base_std = np.sum(np.std(self.last_base10, axis=0))
        done = bool(rot_mat[-1] < 0.5 or np.mean(footz) > -0.1 or np.max(footz) > 0 or
                    (base_std <= 2e-4 and self.steps >= 10) or abs(pose[-1]) > 0.6)
        rewards = 0
        for term in (torso, up, feet, tau, -1 if done else 0, badfoot, footcontact):       # Param_Dict key order :12
            rewards += term
        self.last_basepose = base.copy()
        self.last_base10[1:, :] = self.last_base10[:9, :]
        self.last_base10[0, :] = base
        self.last_foot = fw
        return np.array([torso, up, feet, tau, badfoot, footcontact]), self.reward_p * rewards, done


class SensorStack(object):
    """One robot's observation: the four default sensors of env_builder.py:62-80 in sensor-name order
    (robot_sensors.py:85-162,217-312,314-437,552-578; locomotion_gym_env.py:621-632; env_utils.py:11-42)."""

    def __init__(self, normal=0, motor_dt=13 * 0.002, disp_dt=0.026):
        self.normal, self.motor_dt, self.disp_dt = normal, motor_dt, disp_dt
        self.imu_first = self.motor_first = True
        self.last_angle = np.zeros(12)

    def observe(self, base, rpy, drpy, angles, contact, was_reset, noise=None):
        """`noise`: the 33 Gaussian draws of sensor_mode["noise"] for this observation (env_builder.py:60-71), as INPUTS, in slot
        order displacement 3, rpy 3, drpy 3, motor angles 12, motor rates 12 (robot_sensors.py:281-284, 399-402, 146-148)."""
        base, rpy, angles = np.array(base, np.float64), np.array(rpy, np.float64), np.array(angles, np.float64)
        nz = np.zeros(33) if noise is None else np.asarray(noise, np.float64)
        if was_reset:
            self.imu_first = self.motor_first = True                       # IMUSensor.reset :435-436, MotorAngleAccSensor.reset :159-162
            self.last_angle = np.zeros(12)
            self.cur, self.last, self.yaw_cur, self.yaw_last = base, base, rpy[2], rpy[2]       # on_reset :298-303
        else:
            self.last, self.cur, self.yaw_last, self.yaw_cur = self.cur, base, self.yaw_cur, rpy[2]   # on_step :305-310
        dx, dy, dz = (self.cur - self.last) / self.disp_dt + nz[0:3]       # :280-284 (noise BEFORE the rotation into the local frame)
        disp = np.array([np.cos(self.yaw_last) * dx + np.sin(self.yaw_last) * dy,
                         -np.sin(self.yaw_last) * dx + np.cos(self.yaw_last) * dy, dz])
        if self.normal:
            disp = (disp - np.array([0] * 3)) / np.array([0.1] * 3)
        if self.imu_first:                                                 # :388-390
            self.first_rpy, self.imu_first = rpy.copy(), False
        imu = np.concatenate([rpy - self.first_rpy + nz[3:6], np.asarray(drpy, np.float64) + nz[6:9]])      # :399-402
        if self.normal:
            imu = (imu - np.array([0] * 6)) / np.array([0.1] * 3 + [0.5] * 3)
        if self.motor_first:                                               # :141-145
            acc, self.motor_first = np.zeros(12), False
        else:
            acc = (angles - self.last_angle) / self.motor_dt
        angles, acc = angles + nz[9:21], acc + nz[21:33]                   # :146-148, AFTER the rate was formed ...
        self.last_angle = angles                                           # ... and the NOISY angles are what the next rate starts from (:149)
        motor = np.concatenate((angles, acc))
        if self.normal:
            motor = (motor - np.array([0, 0.9, -1.8] * 4 + [0] * 12)) / np.array([0.1] * 12 + [1] * 12)
        return np.concatenate([disp, np.asarray(contact, np.float64), imu, motor])   # BaseDisplacement < FootContactSensor < IMU < MotorAngleAcc


class ActionFilter(object):
    """robots/action_filter.py:31-99 for one robot: normalised coefficients a, b [12, H + 1], histories newest first."""

    def __init__(self, a, b):
        self.a, self.b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        self.H = self.a.shape[1] - 1
        self.reset()

    def reset(self):                                       # :70-76
        self.xhist = [np.zeros(12) for _ in range(self.H)]
        self.yhist = [np.zeros(12) for _ in range(self.H)]

    def init_history(self, x):                             # :95-99
        self.xhist = [np.array(x, np.float64) for _ in range(self.H)]
        self.yhist = [np.array(x, np.float64) for _ in range(self.H)]

    def filter(self, x):                                   # :78-93
        xs, ys = np.stack(self.xhist, axis=-1), np.stack(self.yhist, axis=-1)
        y = np.multiply(x, self.b[:, 0]) + np.sum(np.multiply(xs, self.b[:, 1:]), axis=-1) - np.sum(np.multiply(ys, self.a[:, 1:]), axis=-1)
        self.xhist = [np.array(x, np.float64)] + self.xhist[:-1]
        self.yhist = [y.copy()] + self.yhist[:-1]
        return y


def foot_positions_in_base_frame(angles):
    """robots/a1.py:105-123,141-147: forward kinematics of the four legs from the 12 motor angles, + HIP_OFFSETS."""
    out = np.zeros((4, 3))
    for i in range(4):
        ab, hip, knee = angles[3 * i], angles[3 * i + 1], angles[3 * i + 2]
        upper = lower = 0.2
        side = 0.08505 * (-1) ** (i + 1)
        leg = np.sqrt(upper**2 + lower**2 + 2 * upper * lower * np.cos(knee))
        swing = hip + knee / 2
        x_hip, z_hip, y_hip = -leg * np.sin(swing), -leg * np.cos(swing), side
        out[i] = np.array([x_hip, np.cos(ab) * y_hip - np.sin(ab) * z_hip, np.sin(ab) * y_hip + np.cos(ab) * z_hip])
    return out + HIP_OFFSETS


class A1Env(object):
    """The composition `A1GymEnv.reset / step` performs (envs/gym_envs/a1_gym_env.py, env_builder.py, MonitorEnv.py:14-25)
    around a physics the caller supplies as recorded world states. One robot. Pinned by tests/golden/a1_env.npz."""

    ETG_MEAN = np.array([2.1505982e-02, 3.6674485e-02, -6.0444288e-02, 2.4625482e-02, 1.5869144e-02, -3.2513142e-02,     # MonitorEnv.py:89-94
                         2.1506395e-02, 3.1869926e-02, -6.0140789e-02, 2.4625063e-02, 1.1628972e-02, -3.2163858e-02])
    ETG_STD = np.array([4.5967497e-02, 2.0340437e-01, 3.7410179e-01, 4.6187632e-02, 1.9441207e-01, 3.9488649e-01,
                        4.5966785e-02, 2.0323379e-01, 3.7382501e-01, 4.6188373e-02, 1.9457331e-01, 3.9302582e-01])

    def __init__(self, w, b, etg=True, normal=0, control_latency=0.002, action_filter=None, segments=None, sensor_mode=None,
                 force_draws=None, dynamics=None):
        # RandomWrapper (MonitorEnv.py:521-662): `force_draws` = the (position, force) pairs its generate_randomforce() would draw
        # (numpy's global stream in the reference; an input here) — None: random_force off. `dynamics` = info["dynamics"] of
        # LocomotionGymEnv.reset (latency s, foot friction, base mass; locomotion_gym_env.py:452-454, MonitorEnv.py:632)
        self.force_draws = None if force_draws is None else iter(force_draws)
        self.dynamics = None if dynamics is None else np.asarray(dynamics, float)
        self.pushes, self.force_info, self.env_steps = [], np.zeros(6), 0
        self.filter = action_filter                                                # Minitaur._BuildActionFilter minitaur.py:1438-1443
        self.path = EtgActionPath(w, b, enabled=etg)
        self.act = A1Actuation(1, control_latency=control_latency)                # POSITION, kp/kd of a1.py:63-68
        self.sensors = SensorStack(normal)
        kw = {} if segments is None else dict(segments=segments)                  # info["env_info"] of the task's terrain
        self.shaping = RewardShaping([1.0, 0.3, 0.2, 0.1, 0.1, 0.1], **kw)        # Param_Dict MonitorEnv.py:12
        self.substeps = 0
        self.etg, self.normal, self.mode = etg, normal, dict(sensor_mode or {})
        rnn = self.mode.get("RNN")
        self.rnn = (rnn["time_steps"], rnn["time_interval"], rnn["mode"]) if rnn and rnn["time_steps"] > 0 else None

    FOOTPOSE_MEAN = np.array([1.7454079e-01, -1.5465108e-01, -2.0661314e-01, 1.7080666e-01, 1.6490668e-01, -2.0865265e-01,   # robot_sensors.py:601-603
                              -1.9902834e-01, -1.2880404e-01, -2.3593837e-01, -2.0215839e-01, 1.3673349e-01, -2.3642859e-01])
    FOOTPOSE_STD = np.array([3.9058894e-02, 2.4757426e-02, 4.2747084e-02, 4.1128017e-02, 2.7591322e-02, 4.3003809e-02,       # :604-606
                             4.3018311e-02, 2.8423777e-02, 4.7990609e-02, 4.6113804e-02, 2.8037265e-02, 4.9409315e-02])

    def select_sensors(self, obs37, inf, world):
        """The sensor list env_builder.py:62-80 builds from sensor_mode, flattened in sensor-NAME order
        (locomotion_gym_env.py:621-632): BaseDisplacement, FootContactSensor | FootForceSensor, FootPoseSensor, IMU,
        MotorAngle | MotorAngleAcc. `obs37` is the default stack (dis, contact, imu 6, motor angle + acceleration)."""
        m = self.mode
        dis, imu, motor, contact, footpose = m.get("dis", 1), m.get("imu", 1), m.get("motor", 1), m.get("contact", 1), m.get("footpose", 0)
        if (dis, imu, motor, contact, bool(footpose)) == (1, 1, 1, 1, False):
            return obs37
        parts = []
        if dis:
            parts.append(obs37[0:3])
        if contact == 1:
            parts.append(obs37[3:7])
        elif contact == 2:                                   # SimpleFootForceSensor robot_sensors.py:546-548: flags + |normal force| / 100
            parts.append(np.asarray(world["force"], float))
        if footpose:                                         # FootPoseSensor :607-611
            fp = np.asarray(inf["footposition"]).reshape(-1)
            parts.append((fp - self.FOOTPOSE_MEAN) / self.FOOTPOSE_STD if self.normal else fp)
        if imu == 1:
            parts.append(obs37[7:13])
        elif imu == 2:                                       # IMUSensor(channels dR dP dY), built WITHOUT `normal` (env_builder.py:67)
            parts.append(np.asarray(inf["drpy"], float))
        if motor == 1:
            parts.append(obs37[13:37])
        elif motor == 2:                                     # MotorAngleSensor :74-84 (no normalisation)
            parts.append(np.asarray(inf["joint_angle"], float))
        return np.concatenate(parts) if parts else np.zeros(0)

    def wrap_observation(self, obs, yaw, etg_obs, d_yaw, on_reset):
        """ObservationWrapper.reset :136-179 / step :181-221."""
        if self.etg and self.mode.get("ETG"):
            out = self.path.last_etg_act
            if self.normal:
                out = (out - self.ETG_MEAN) / self.ETG_STD
            obs = np.concatenate((obs, out), axis=0)
        if self.etg and self.mode.get("ETG_obs"):
            obs = np.concatenate((obs, etg_obs), axis=0)
        if self.mode.get("force_vec"):                                              # :150-152 / :194-196
            obs = np.concatenate((obs, self.force_info), axis=0)
        if self.mode.get("dynamic_vec"):                                            # :154-156 / :198-200
            obs = np.concatenate((obs, self.dynamics), axis=0)
        if self.mode.get("yaw"):
            obs = np.concatenate((obs, np.array([np.cos(d_yaw - yaw), np.sin(d_yaw - yaw)])), axis=0)
        if self.rnn:
            steps, interval, mode = self.rnn
            if on_reset:
                self.obs_history = np.zeros((steps * interval, obs.shape[0]))
            frames = [self.obs_history[t * interval].copy() for t in range(steps)] + [obs.copy()]
            if not on_reset:
                self.obs_history[:-1] = self.obs_history[1:].copy()
            self.obs_history[-1] = obs
            obs = np.stack(frames, axis=0) if mode == "GRU" else np.array(frames).reshape(-1)
        return obs

    def time_since_reset(self):
        return self.substeps * 0.002                                               # GetTimeSinceReset minitaur.py:228-230

    def _push(self, new):
        """RandomWrapper: applyExternalForce(base, force, position, LINK_FRAME) — it acts during the NEXT stepSimulation only."""
        if new:
            self.force_pos, self.force_vec = [np.asarray(x, float) for x in next(self.force_draws)]
        self.pushes.append((self.total_substeps, self.force_vec.copy(), self.force_pos.copy()))
        self.force_info = np.concatenate((self.force_pos / np.array([0.2, 0.05, 0.05]), self.force_vec / 50))

    def _random_force_after_step(self):
        """RandomWrapper.step :644-660, after the inner env.step (env_step_counter already incremented)."""
        self.force_info = np.zeros(6)
        if self.force_draws is None:
            return
        c = self.env_steps
        if c % 100 == 0:
            self._push(True)
        elif c % 100 < 50:
            self._push(False)

    def robot_step(self, command, true_obs):
        """Minitaur.Step with the world's 13 recorded sub-step states; returns the 13 x 12 torques."""
        torques = []
        if self.filter is not None:                                                # _FilterAction minitaur.py:1448-1457
            if self.substeps == 0:
                self.filter.init_history(self.act.sensors()[0][0])
            command = self.filter.filter(command)
        for i in range(13):
            torques.append(self.act.apply_action(self.act.process_action(command[None], i))[0])
            t = true_obs[i]
            self.act.receive_observation(t[None, 0:12], t[None, 12:24], t[None, 36:40], t[None, 40:43])
            self.substeps += 1
        self.act.last_action = command[None]
        return np.array(torques)

    def info(self, world):
        """What LocomotionGymEnv.step / reset put in `info` (locomotion_gym_env.py:440-455,534-545) that is computed in
        Python: foot positions (FK of the delayed motor angles), energy, drpy; base / pose / rot_mat / contacts are the world's."""
        ang, vel, tor, rate, energy = self.act.sensors()
        return dict(footposition=foot_positions_in_base_frame(ang[0]), joint_angle=ang[0], drpy=rate[0], energy=energy[0])

    @staticmethod
    def reset_pose(add_height, yaw=0.0, add_x=0.0):
        """What LocomotionGymEnv.reset hands Minitaur.Reset (locomotion_gym_env.py:327-338) and that hands
        resetBasePositionAndOrientation (minitaur.py:426-429): position [add_x, 0, 0.28 + add_height], quaternion (x, y, z, w) of a
        rotation by `yaw` about z. `add_x` is the U(-0.2, 0.1) draw of x_noise (numpy's global stream: an input)."""
        return [0 + add_x, 0, 0.28 + add_height], [0, 0, np.sin(yaw / 2.0), np.cos(yaw / 2.0)]

    def reset(self, reset_true_obs, reset_world, hidden_true_obs, hidden_world, d_yaw=0, ETG_w=None, ETG_b=None, segments=None):
        """A1GymEnv.reset(**kwargs): LocomotionGymEnv.reset (a `hardset` terrain's env_info arrives as `segments`,
        locomotion_gym_env.py:297-301; robot.Reset: history cleared, one observation; sensors reset), ETGWrapper.reset (new
        parameters from ETG_w / ETG_b first, MonitorEnv.py:250-253), then RewardShaping.reset's hidden zero-action step (:305-318).
        Returns (the hidden step's command, torques, the observation reset() returns)."""
        if segments is not None:
            self.shaping.segments = [tuple(s) for s in segments]
        if ETG_w is not None:
            self.path.w = np.asarray(ETG_w)
        if ETG_b is not None:
            self.path.b = np.asarray(ETG_b)
        self.act.reset(); self.substeps = 0
        self.env_steps = 0                                                         # LocomotionGymEnv._env_step_counter :414
        self.total_substeps = getattr(self, "total_substeps", 0)
        if self.filter is not None:
            self.filter.reset()                                                    # _ResetActionFilter (Minitaur.Reset :443-444)
        t = reset_true_obs
        self.act.receive_observation(t[None, 0:12], t[None, 12:24], t[None, 36:40], t[None, 40:43])
        inf = self.info(reset_world)
        obs0 = self.sensors.observe(reset_world["base"], reset_world["pose"], inf["drpy"], inf["joint_angle"], reset_world["contact"], True)
        obs0 = self.select_sensors(obs0, inf, reset_world)
        etg_obs0 = self.path.reset(self.time_since_reset())
        self.force_info = np.zeros(6)
        if self.force_draws is not None:                                           # RandomWrapper.reset :634-640: a fresh force, applied
            self._push(True)
        self.wrap_observation(obs0, reset_world["pose"][-1], etg_obs0, d_yaw, True)
        cmd, torques, obs, _ = self._step(np.zeros(12), hidden_true_obs, hidden_world, shaped=False)
        self.shaping.reset(reset_world["base"], reset_world["rot_mat"], inf["footposition"])
        return cmd, torques, obs

    def _step(self, action, true_obs, world, shaped=True, d_yaw=0):
        cmd, etg_obs = self.path.step(action, self.time_since_reset())
        torques = self.robot_step(cmd, true_obs)
        self.env_steps += 1
        self.total_substeps += 13
        self._random_force_after_step()
        inf = self.info(world)
        obs = self.sensors.observe(world["base"], world["pose"], inf["drpy"], inf["joint_angle"], world["contact"], False)
        obs = self.select_sensors(obs, inf, world)
        obs = self.wrap_observation(obs, world["pose"][-1], etg_obs, d_yaw, False)
        out = None
        if shaped:
            out = self.shaping.step(world["base"], world["pose"], world["rot_mat"], inf["footposition"], world["contact"],
                                    inf["energy"], world["bad"], d_yaw)
        return cmd, torques, obs, (out, inf)

    def step(self, action, true_obs, world, d_yaw=0):
        return self._step(action, true_obs, world, d_yaw=d_yaw)
