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
