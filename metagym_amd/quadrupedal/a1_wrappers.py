"""The control-side wrappers of `A1GymEnv.step` (metagym/quadrupedal/envs/env_wrappers/MonitorEnv.py), batched:

    EtgActionPath   ETGWrapper (:222-273) + TrajectoryGeneratorWrapperEnv + LaikagoPoseOffsetGenerator:
                    policy action [N,12] -> motor command [N,12] (what LocomotionGymEnv.step / robot.Step receive)
    RewardShaping   RewardShaping (:275-519): the step's `info` -> reward terms, reward, done

Both run as one HIP kernel per call behind mg_a1_etg_action / mg_a1_reward_* (metagym_amd/csrc/a1.hip). float64
tensors `[num_envs, k]` on the device. The physics between them is the caller's (see a1_actuators.py)."""
import ctypes as C

import numpy as np
import torch

from .. import _lib

Param_Dict = {'torso': 1.0, 'up': 0.3, 'feet': 0.2, 'tau': 0.1, 'done': 1, 'velx': 0, 'badfoot': 0.1, 'footcontact': 0.1}   # MonitorEnv.py:12
FLAT_GROUND = [[-100, 100, np.array([1, 0, 0, 0, 0, 0, 0])]]          # locomotion_gym_env.py:76 (env_info)


def _u8(mask, device):
    """A bool / uint8 mask as the contiguous uint8 array the kernels read (a contiguous bool tensor IS one: reinterpreted, no copy)."""
    m = torch.as_tensor(mask, device=device)
    if m.dtype == torch.bool and m.is_contiguous():
        return m.view(torch.uint8)
    return m.to(torch.uint8).contiguous()


def _soa(x, n, k, device, dtype=torch.float64):
    x = torch.as_tensor(x, dtype=dtype, device=device)
    assert x.shape == (n, k), "expected [num_envs, %d], got %s" % (k, tuple(x.shape))
    return x.t().contiguous()


class EtgActionPath(object):
    def __init__(self, num_envs, device="cuda:0", ETG=1, ETG_T=0.5, ETG_T2=0.5, ETG_H=20, ETG_w=None, ETG_b=None, act_mode="traj",
                 task_mode="normal", action_space=0):
        self.device = _lib.canonical_device(device)
        if self.device.type != "cuda":
            raise _lib.MetaGymHipError("metagym_amd has no CPU path: device must be a ROCm GPU, got %r" % (device,))
        assert act_mode in ("traj", "pose") and 1 <= ETG_H <= _lib.A1_ETG_MAX_H
        self._lib = _lib.load()
        self.num_envs, self.H = int(num_envs), int(ETG_H)
        c = self._cfg = _lib.A1EtgConfig()
        c.enabled, c.H, c.T, c.T2_ratio, c.sigma_sq, c.amp = int(bool(ETG)), self.H, ETG_T, ETG_T2, 0.04, 0.2      # MonitorEnv.py:238
        phase = np.array([-np.pi / 2, 0])                                                                      # :236
        c.phase[:] = list(phase)
        c.omega = 2.0 * np.pi / ETG_T                                                                          # ETG_model.py:20
        for h in range(self.H):                                                                                # :22-25
            t_now = h * ETG_T / (self.H - 0.9)
            c.u[h][:] = [float(c.amp * np.sin(phase[i] + t_now * c.omega)) for i in range(2)]
        self.set_etg_parameters(np.zeros((3, self.H)) if ETG_w is None else ETG_w, np.zeros(3) if ETG_b is None else ETG_b)
        c.act_mode_pose, c.gallop, c.etg_weight, c.action_space = int(act_mode == "pose"), int(task_mode == "gallop"), 1.0, int(action_space)
        c.pose[:] = [0.0, 0.9, -1.8] * 4                                                                       # laikago_pose_utils.py:17-19
        self.last_ETG_act = torch.zeros(12, self.num_envs, dtype=torch.float64, device=self.device)

    def set_etg_parameters(self, w, b):
        """ETG_w [3, H], ETG_b [3] (MonitorEnv.py:240-245, or reset(ETG_w=, ETG_b=) :250-253). They travel by value with every
        launch: a step captured into a hipGraph keeps the parameters it was captured with."""
        w, b = np.asarray(w, dtype=np.float64), np.asarray(b, dtype=np.float64)
        assert w.shape == (3, self.H) and b.shape == (3,)
        for a in range(3):
            self._cfg.w[a][:self.H] = list(w[a])
        self._cfg.b[:] = list(b)

    def etg_w(self):
        return np.asarray([[self._cfg.w[a][h] for h in range(self.H)] for a in range(3)])

    def etg_b(self):
        return np.asarray([self._cfg.b[a] for a in range(3)])

    def _launch(self, action, t, command, etg_obs, act=None):
        tt = torch.as_tensor(t, dtype=torch.float64, device=self.device)
        if tt.dim() == 0:
            tt = tt.expand(self.num_envs)
        tt = tt.contiguous()
        with torch.cuda.device(self.device):
            rc = self._lib.mg_a1_etg_action(C.byref(self._cfg), self.num_envs, _lib.ptr(self.last_ETG_act if act is None else act), _lib.ptr(action),
                                            _lib.ptr(tt), _lib.ptr(command), _lib.ptr(etg_obs), _lib.current_stream(self.device))
        _lib.check(rc, "mg_a1_etg_action")

    def reset(self, t=0.0, mask=None):
        """ETGWrapper.reset (:246-259): returns info["ETG_obs"] `[num_envs, H]` (None when the ETG is off). `mask`: only these
        robots' ETG state is refreshed."""
        if not self._cfg.enabled:
            return None
        obs = torch.empty(self.H, self.num_envs, dtype=torch.float64, device=self.device)
        if mask is None:
            self._launch(None, t, None, obs)
        else:       # the reset output of every robot goes to a scratch array (a reset reads nothing of the previous output); the masked
            #         robots' columns are then taken over: two launches
            if not hasattr(self, "_reset_scratch"):
                self._reset_scratch = torch.empty_like(self.last_ETG_act)
            self._launch(None, t, None, obs, act=self._reset_scratch)
            torch.where(torch.as_tensor(mask, device=self.device).bool(), self._reset_scratch, self.last_ETG_act, out=self.last_ETG_act)
        return obs.t()

    def step(self, action, t):
        """action `[num_envs, 12]`, t = time since reset before this env step (scalar or `[num_envs]`).
        Returns (motor command `[num_envs, 12]`, info["ETG_obs"] or None); info["ETG_act"] is `last_ETG_act.t()`."""
        a = _soa(action, self.num_envs, 12, self.device)
        cmd = torch.empty(12, self.num_envs, dtype=torch.float64, device=self.device)
        obs = torch.empty(self.H, self.num_envs, dtype=torch.float64, device=self.device) if self._cfg.enabled else None
        self._launch(a, t, cmd, obs)
        return cmd.t(), (None if obs is None else obs.t())


class RewardShaping(object):
    def __init__(self, num_envs, device="cuda:0", param=Param_Dict, reward_p=1, vel_d=0.6, vel_mode="max", env_info=FLAT_GROUND):
        self.device = _lib.canonical_device(device)
        if self.device.type != "cuda":
            raise _lib.MetaGymHipError("metagym_amd has no CPU path: device must be a ROCm GPU, got %r" % (device,))
        if vel_mode not in ("max", "equal"):       # MonitorEnv.py:512-518 knows these two
            raise ValueError("vel_mode %r: the reference has 'max' and 'equal'" % (vel_mode,))
        self._lib = _lib.load()
        self.num_envs = N = int(num_envs)
        c = self._cfg = _lib.A1RewardConfig()
        c.w_torso, c.w_up, c.w_feet, c.w_tau = param['torso'], param['up'], param['feet'], param['tau']
        c.w_badfoot, c.w_footcontact, c.reward_p, c.vel_d = param['badfoot'], param['footcontact'], reward_p, vel_d
        c.vel_mode = 1 if vel_mode == "equal" else 0
        c.cw_half = float(np.arctanh(np.sqrt(0.95)) / 0.5)          # c_prec's w (:421-425) for m = 0.5 and m = 0.4
        c.cw_04 = float(np.arctanh(np.sqrt(0.95)) / 0.4)
        assert len(env_info) <= _lib.A1_MAX_SEGMENTS
        c.n_segments = len(env_info)
        for i, (x0, x1, vec) in enumerate(env_info):
            c.seg[i][:] = [float(x0), float(x1), float(vec[0]), float(vec[1]), float(vec[4])]
        f64 = dict(dtype=torch.float64, device=self.device)
        self._t = dict(last_base=torch.zeros(3, N, **f64), last_base10=torch.zeros(30, N, **f64), last_foot=torch.zeros(12, N, **f64),
                       vd2=torch.zeros(2, N, **f64), steps=torch.zeros(N, dtype=torch.int32, device=self.device))
        s = self._st = _lib.A1RewardState()
        for k, t in self._t.items():
            setattr(s, k, t.data_ptr())

    @staticmethod
    def pack_env_info(env_info):
        """info["env_info"] (a list of [x0, x1, env_vec[7]]) as the [MG_A1_MAX_SEGMENTS][5] rows the kernels read + its length."""
        assert len(env_info) <= _lib.A1_MAX_SEGMENTS, "%d terrain stretches (max %d)" % (len(env_info), _lib.A1_MAX_SEGMENTS)
        rows = np.zeros((_lib.A1_MAX_SEGMENTS, 5))
        for i, (x0, x1, vec) in enumerate(env_info):
            rows[i] = [float(x0), float(x1), float(vec[0]), float(vec[1]), float(vec[4])]
        return rows, len(env_info)

    def set_terrain_table(self, seg_table, seg_count, terrain_id):
        """Per-robot env_info (mg_a1_reward_config.seg_table): device tensors f64 [T, MG_A1_MAX_SEGMENTS, 5], i32 [T], i32 [N] —
        robot e looks its stretches up in course terrain_id[e]. The tensors are read at launch time (A1GymEnv rewrites them in
        place when a reset builds a new course)."""
        assert seg_table.dtype == torch.float64 and seg_table.shape[1:] == (_lib.A1_MAX_SEGMENTS, 5) and seg_table.is_contiguous()
        assert seg_count.dtype == torch.int32 and seg_count.shape == (seg_table.shape[0],)
        assert terrain_id.dtype == torch.int32 and terrain_id.shape == (self.num_envs,)
        self._seg_table, self._seg_count, self._terrain_id = seg_table, seg_count, terrain_id
        c = self._cfg
        c.seg_table, c.seg_count, c.terrain_id = seg_table.data_ptr(), seg_count.data_ptr(), terrain_id.data_ptr()

    def reset(self, base, rot_mat, footposition, mask=None):
        """RewardShaping.reset (:305-318) with the RESET info's base [N,3], rot_mat [N,9], footposition [N,12] (base frame)."""
        N, d = self.num_envs, self.device
        m = None if mask is None else _u8(mask, d)
        # (named locals: the SoA copies must outlive the launch call, _lib.ptr() only keeps their addresses)
        b, r, f = _soa(base, N, 3, d), _soa(rot_mat, N, 9, d), _soa(footposition, N, 12, d)
        with torch.cuda.device(d):
            rc = self._lib.mg_a1_reward_reset(C.byref(self._cfg), N, C.byref(self._st), _lib.ptr(b), _lib.ptr(r), _lib.ptr(f),
                                              _lib.ptr(m), _lib.current_stream(d))
        _lib.check(rc, "mg_a1_reward_reset")

    def step(self, base, pose, rot_mat, footposition, real_contact, energy, bad_foot_contacts, d_yaw=None):
        """-> (reward [N], done [N] bool, terms dict of [N]) from this step's info (locomotion_gym_env.py:534-545)."""
        N, d = self.num_envs, self.device
        f64 = dict(dtype=torch.float64, device=d)
        terms, reward = torch.empty(6, N, **f64), torch.empty(N, **f64)
        done = torch.empty(N, dtype=torch.uint8, device=d)
        en = torch.as_tensor(energy, **f64).contiguous()
        bad = torch.as_tensor(bad_foot_contacts, device=d).to(torch.int32).contiguous()
        dy = None if d_yaw is None else torch.as_tensor(d_yaw, **f64).expand(N).contiguous()
        b, p, r = _soa(base, N, 3, d), _soa(pose, N, 3, d), _soa(rot_mat, N, 9, d)
        f, ct = _soa(footposition, N, 12, d), _soa(real_contact, N, 4, d)
        with torch.cuda.device(d):
            rc = self._lib.mg_a1_reward_step(C.byref(self._cfg), N, C.byref(self._st), _lib.ptr(b), _lib.ptr(p), _lib.ptr(r),
                                             _lib.ptr(f), _lib.ptr(ct), _lib.ptr(en), _lib.ptr(bad), _lib.ptr(dy),
                                             _lib.ptr(terms), _lib.ptr(reward), _lib.ptr(done), _lib.current_stream(d))
        _lib.check(rc, "mg_a1_reward_step")
        names = ("torso", "up", "feet", "tau", "badfoot", "footcontact")
        return reward, done.view(torch.bool), {k: terms[i] for i, k in enumerate(names)}     # (0 / 1 bytes: reinterpreted)

    def state_dict(self):
        return {k: t.clone() for k, t in self._t.items()}

    def load_state_dict(self, sd):
        for k, t in self._t.items():
            assert sd[k].shape == t.shape, k
            t.copy_(sd[k])


class SensorStack(object):
    """The observation of `A1GymEnv` (37 float64): BaseDisplacementSensor (local frame), FootContactSensor, IMUSensor
    (R P Y dR dP dY), MotorAngleAccSensor in sensor-name order (envs/sensors/robot_sensors.py, locomotion_gym_env.py:621-632)."""

    NOISE_SIGMA = [1e-2] * 3 + [6e-2] * 3 + [1e-1] * 3 + [1e-2] * 12 + [0.5] * 12     # robot_sensors.py:281-284, 399-402, 146-148

    def __init__(self, num_envs, device="cuda:0", normal=0, num_action_repeat=13, sim_time_step=0.002, noise=False, seed=0,
                 noise_source=None):
        """`noise` = sensor_mode["noise"] (env_builder.py:60-71): every observation adds Gaussian draws inside the sensors — 33 per
        robot (NOISE_SIGMA). The reference takes them from numpy's global stream; here a device generator seeded by `seed`, or
        `noise_source()` -> [33, N] already-scaled values (tests replay the reference's own draws through it)."""
        self.noise, self._noise_source = bool(noise), noise_source
        self.device = _lib.canonical_device(device)
        if self.device.type != "cuda":
            raise _lib.MetaGymHipError("metagym_amd has no CPU path: device must be a ROCm GPU, got %r" % (device,))
        self._lib = _lib.load()
        self.num_envs = N = int(num_envs)
        c = self._cfg = _lib.A1SensorConfig()
        c.normal, c.disp_dt, c.motor_dt = int(normal), 0.026, num_action_repeat * sim_time_step        # env_builder.py:49
        f64 = dict(dtype=torch.float64, device=self.device)
        self._t = dict(base_last=torch.zeros(3, N, **f64), base_cur=torch.zeros(3, N, **f64), yaw=torch.zeros(2, N, **f64),
                       first_rpy=torch.zeros(3, N, **f64), last_angle=torch.zeros(12, N, **f64),
                       first=torch.full((N,), 3, dtype=torch.int32, device=self.device))
        s = self._st = _lib.A1SensorState()
        for k, t in self._t.items():
            setattr(s, k, t.data_ptr())
        if self.noise:
            self._noise = torch.zeros(33, N, **f64)               # persistent: a captured step keeps reading this buffer
            self._sigma = torch.tensor(self.NOISE_SIGMA, **f64).reshape(33, 1)
            self.noise_gen = torch.Generator(device=self.device)
            self.noise_gen.manual_seed(int(seed) + 0x5e5)
            s.noise = self._noise.data_ptr()

    def observe(self, base_position, base_rpy, base_rpy_rate, motor_angles, foot_contacts, reset_mask=None):
        """-> obs `[num_envs, 37]`. `reset_mask`: robots that were just reset (sensor.reset() + on_reset instead of on_step)."""
        N, d = self.num_envs, self.device
        b, r, dr = _soa(base_position, N, 3, d), _soa(base_rpy, N, 3, d), _soa(base_rpy_rate, N, 3, d)
        a, ct = _soa(motor_angles, N, 12, d), _soa(foot_contacts, N, 4, d)
        m = None if reset_mask is None else _u8(reset_mask, d)
        obs = torch.empty(N, _lib.A1_SENSOR_OBS_DIM, dtype=torch.float64, device=d)
        if self.noise:
            if self._noise_source is not None:
                self._noise.copy_(torch.as_tensor(self._noise_source(), dtype=torch.float64, device=d).reshape(33, -1).expand(33, N))
            else:
                self._noise.copy_(torch.randn(33, N, generator=self.noise_gen, dtype=torch.float64, device=d) * self._sigma)
        with torch.cuda.device(d):
            rc = self._lib.mg_a1_observation(C.byref(self._cfg), N, C.byref(self._st), _lib.ptr(b), _lib.ptr(r), _lib.ptr(dr),
                                             _lib.ptr(a), _lib.ptr(ct), _lib.ptr(m), _lib.ptr(obs), _lib.current_stream(d))
        _lib.check(rc, "mg_a1_observation")
        return obs


class ActionFilter(object):
    """`ActionFilter` / `ActionFilterButter` / `ActionFilterExp` of quadrupedal/robots/action_filter.py for N robots.
    Give normalised coefficient arrays `a`, `b` `[12, hist_len + 1]`, or use `butter()` for the reference's default
    (`Minitaur._BuildActionFilter`, minitaur.py:1438-1443: 2nd-order Butterworth low-pass at 4 Hz)."""

    def __init__(self, num_envs, a, b, device="cuda:0"):
        self.device = _lib.canonical_device(device)
        if self.device.type != "cuda":
            raise _lib.MetaGymHipError("metagym_amd has no CPU path: device must be a ROCm GPU, got %r" % (device,))
        self._lib = _lib.load()
        self.num_envs = N = int(num_envs)
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        assert a.shape == b.shape and a.shape[0] == 12 and 2 <= a.shape[1] <= _lib.A1_FILTER_MAX_HIST + 1
        c = self._cfg = _lib.A1FilterConfig()
        c.hist_len = H = a.shape[1] - 1
        for j in range(12):
            c.a[j][:H + 1] = list(a[j] / a[j][0])          # action_filter.py:55-57
            c.b[j][:H + 1] = list(b[j] / a[j][0])
        self.xhist = torch.zeros(H, 12, N, dtype=torch.float64, device=self.device)
        self.yhist = torch.zeros(H, 12, N, dtype=torch.float64, device=self.device)

    @classmethod
    def butter(cls, num_envs, sampling_rate, device="cuda:0", lowcut=(0.0,), highcut=(4.0,), order=2):
        """ActionFilterButter (action_filter.py:102-185); the coefficients are scipy.signal.butter's, like the reference's."""
        from scipy.signal import butter
        lowcut, highcut = [float(x) for x in lowcut], [float(x) for x in highcut]
        a, b = [], []
        for l, h in zip(lowcut, highcut):
            nyq = 0.5 * sampling_rate
            bb, aa = butter(order, [l / nyq, h / nyq], btype="band") if l else butter(order, [h / nyq], btype="low")
            a.append(aa); b.append(bb)
        if len(a) == 1:
            a, b = a * 12, b * 12
        return cls(num_envs, np.stack(a), np.stack(b), device)

    def _call(self, x, y, mask, mode):
        m = None if mask is None else torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
        with torch.cuda.device(self.device):
            rc = self._lib.mg_a1_action_filter(C.byref(self._cfg), self.num_envs, _lib.ptr(self.xhist), _lib.ptr(self.yhist),
                                               _lib.ptr(x), _lib.ptr(y), _lib.ptr(m), mode, _lib.current_stream(self.device))
        _lib.check(rc, "mg_a1_action_filter")

    def reset(self, mask=None):
        self._call(None, None, mask, 1)

    def filter(self, x, init_mask=None):
        """x `[num_envs, 12]` -> filtered `[num_envs, 12]`. Robots in `init_mask` first get `init_history(x)` (the reference
        does that with the current motor angles on the first step of an episode, minitaur.py:1452-1454 — pass those as x
        through `init_history` instead when they differ from the command)."""
        xs = _soa(x, self.num_envs, 12, self.device)
        y = torch.empty_like(xs)
        self._call(xs, y, init_mask, 0)
        return y.t()

    def init_history(self, x, mask=None):
        """ActionFilter.init_history (action_filter.py:95-99) for the robots in `mask` (None = all)."""
        xs = _soa(x, self.num_envs, 12, self.device)
        if mask is None:
            self.xhist[:] = xs
            self.yhist[:] = xs
        else:                           # torch.where, not boolean indexing: no host sync
            m = torch.as_tensor(mask, device=self.device).bool()
            self.xhist.copy_(torch.where(m, xs, self.xhist))
            self.yhist.copy_(torch.where(m, xs, self.yhist))
