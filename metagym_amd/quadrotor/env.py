"""Batched `Quadrotor` env — host-side mirror of metagym/quadrotor/env.py:30 (`Quadrotor(gym.Env)`).

Same constructor arguments, `reset()` / `step()` / `action_space` / `observation_space` names and
the same 16-entry observation order, but every call advances `num_envs` independent quadrotors
with ONE launch of the hand-written gfx950 kernel behind `mg_quadrotor_step` (include/metagym_hip.h,
metagym_amd/csrc/quadrotor.hip) and returns torch-ROCm tensors of shape [num_envs, ...].

Differences forced by batching (documented in DESIGN.md):
  * `_check_failure` (quadrotorsim.py:212-221) cannot raise per lane: `info['failed']` is a uint8
    tensor (0 ok, 1/2/3 = range/velocity/body-rate), a failed env reports done=1, reward=0.
  * the reference draws reset noise from numpy's global RNG (quadrotorsim.py:243-251); here each
    env owns the draws `RandomState(seed).random_sample((num_envs, 4, 3))[e]`, so env 0 reproduces
    `np.random.seed(seed); env.reset()` of the reference exactly.
  * `info` values are float32 views of the observation (the reference hands back numpy scalars).
"""
import ctypes as C
import json
import os
from collections.abc import Mapping

import numpy as np
import torch

from .. import _lib
from ..spaces import Box, Space

TASKS = {"no_collision": 0, "velocity_control": 1, "hovering_control": 2}

# torch's current HIP stream of a device as a raw handle (an int): the private C-level query when this torch has it, else the
# public route through a Stream object (same value; a side stream entered for a hipGraph capture is seen by both)
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or (lambda index: torch.cuda.current_stream(index).cuda_stream)

# Physical constants of the reference's metagym/quadrotor/config.json (same JSON schema; pass
# `simulator_conf=<path>` to load another file with that schema).
DEFAULT_SIM_CONFIG = {
    "precision": 0.001, "quality": 0.5,
    "inertia": {"xx": 0.0135, "xy": 0.0, "xz": 0.0, "yy": 0.0135, "yz": 0.0, "zz": 0.024},
    "drag": {"m_xx": 0.074, "m_yy": 0.074, "m_zz": 0.0506, "f_xx": 0.12, "f_yy": 0.12, "f_zz": 0.10},
    "gravity_center": {"x": 0.0, "y": 0.0, "z": 0.0},
    "thrust": {"CT": ["1.538e-5", "-2.5e-4", "0.0"], "Mm": "0.010", "Jm": "2.573e-4", "RA": "0.2010",
               "phi": "0.017242179827506"},
    "propeller": [{"x": 0.18, "y": 0.18, "z": 0.0}, {"x": -0.18, "y": 0.18, "z": 0.0},
                  {"x": -0.18, "y": -0.18, "z": 0.0}, {"x": 0.18, "y": -0.18, "z": 0.0}],
    "fail": {"velocity": 100.0, "w": 1000.0, "range": 1000.0},
    "electric": {"min_voltage": 0.10, "max_voltage": 15.0},
    "init_velocity": {"x": 0, "y": 0, "z": 0, "noisy": 2.0},
    "init_angular_velocity": {"x": 0, "y": 0, "z": 0, "noisy": 5.0},
}

OBS_KEYS = ["b_v_x", "b_v_y", "b_v_z", "b_x", "b_y", "b_z", "acc_x", "acc_y", "acc_z",
            "gyro_x", "gyro_y", "gyro_z", "pitch", "roll", "yaw", "z",           # env.py:78-84,193-196
            "next_target_g_v_x", "next_target_g_v_y", "next_target_g_v_z"]       # velocity_control only, env.py:85-86


def _fill_config(cfg, sim, dt, nt, task, healthy_reward):
    """quadrotorsim.py:50-109 `_parse_cfg`: python floats stay doubles, matrices are float32."""
    cfg.precision = float(sim["precision"])
    cfg.quality = float(sim["quality"])
    i = sim["inertia"]
    inertia = [i["xx"], i["xy"], i["xz"], i["xy"], i["yy"], i["yz"], i["xz"], i["yz"], i["zz"]]
    for k in range(9):
        cfg.inertia[k] = float(inertia[k])
        cfg.drag_m[k] = 0.0
        cfg.drag_f[k] = 0.0
    d = sim["drag"]
    cfg.drag_m[0], cfg.drag_m[4], cfg.drag_m[8] = float(d["m_xx"]), float(d["m_yy"]), float(d["m_zz"])
    cfg.drag_f[0], cfg.drag_f[4], cfg.drag_f[8] = float(d["f_xx"]), float(d["f_yy"]), float(d["f_zz"])
    g = sim["gravity_center"]
    cfg.gravity_center[0], cfg.gravity_center[1], cfg.gravity_center[2] = float(g["x"]), float(g["y"]), float(g["z"])
    t = sim["thrust"]
    cfg.ct0, cfg.ct1, cfg.ct2 = float(t["CT"][0]), float(t["CT"][1]), float(t["CT"][2])
    cfg.mm, cfg.jm, cfg.phi, cfg.ra = float(t["Mm"]), float(t["Jm"]), float(t["phi"]), float(t["RA"])
    f = sim["fail"]
    cfg.fail_velocity, cfg.fail_range, cfg.fail_w = float(f["velocity"]), float(f["range"]), float(f["w"])
    for p in range(4):
        for a, ax in enumerate("xyz"):
            cfg.prop_coord[3 * p + a] = float(sim["propeller"][p][ax])
    cfg.max_voltage = float(sim["electric"]["max_voltage"])
    cfg.min_voltage = float(sim["electric"]["min_voltage"])
    cfg.dt = float(dt)
    cfg.nt = int(nt)
    cfg.task = TASKS[task]
    cfg.healthy_reward = float(healthy_reward)


class _LazyInfo(Mapping):
    """`info` dict of the reference (env.py:163-164: the 16 state keys, 'z' including z_offset) as
    lazily-created [N] views of the observation tensor, plus 'failed'. Building 17 tensor views
    eagerly would cost more host time per step than the whole simulation kernel."""

    def __init__(self, obs, failed):
        self._obs, self._failed = obs, failed

    def __getitem__(self, k):
        if k == "failed":
            return self._failed
        i = OBS_KEYS.index(k)
        if i >= self._obs.shape[1]:
            raise KeyError(k)
        return self._obs[:, i]

    def __iter__(self):
        return iter(OBS_KEYS[:self._obs.shape[1]] + ["failed"])

    def __len__(self):
        return self._obs.shape[1] + 1


class Quadrotor(object):
    """`num_envs` quadrotors stepped in lock-step on one MI355X.

    Args (after num_envs/device identical to the reference, env.py:46-54):
        num_envs (int): number of parallel environments N.
        device: torch device with the HIP runtime ('cuda', 'cuda:0', ...).
        dt, nt, seed, task, map_file, simulator_conf, healthy_reward: as in the reference.
    """

    metadata = {"render.modes": []}

    def __init__(self, num_envs=1, device="cuda", dt=0.01, nt=1000, seed=0, task="no_collision",
                 map_file=None, simulator_conf=None, healthy_reward=1.0, auto_reset=False, env_id_base=0,
                 exact_reward=True, copy_outputs=False, **kwargs):
        assert task in TASKS, "Invalid task setting"
        self._lib = _lib.load()
        self.num_envs = int(num_envs)
        self.device = _lib.canonical_device(device)
        if self.device.type != "cuda":
            raise _lib.MetaGymHipError("metagym_amd runs on an AMD GPU only (got device %r); there "
                                       "is no CPU path" % (device,))
        self.dt, self.nt, self.task, self.healthy_reward = dt, nt, task, healthy_reward
        # step() / reset() hand back the SAME output tensors every call (the kernel writes into them; a hipGraph replays on
        # them): keep `obs` across steps only as `obs.clone()`. `copy_outputs=True` returns fresh tensors like the reference's
        # fresh numpy arrays, at the price of four device copies per step.
        self.copy_outputs = bool(copy_outputs)
        if simulator_conf is None:
            self.sim_config = json.loads(json.dumps(DEFAULT_SIM_CONFIG))
        else:
            assert os.path.exists(simulator_conf), "Simulator config file does not exist"
            with open(simulator_conf, "r") as f:
                self.sim_config = json.load(f)

        N, dev = self.num_envs, self.device
        # --- per-env state, SoA, owned by torch (116 B/env) ---------------------------------
        self.pos = torch.zeros(3, N, dtype=torch.float32, device=dev)
        self.vel = torch.zeros(3, N, dtype=torch.float64, device=dev)
        self.omega = torch.zeros(3, N, dtype=torch.float64, device=dev)
        self.propw = torch.zeros(4, N, dtype=torch.float32, device=dev)
        self.rot = torch.zeros(9, N, dtype=torch.float32, device=dev)
        self.rot[0::4] = 1.0
        self.ct = torch.zeros(N, dtype=torch.int32, device=dev)
        # auto-resets each env has gone through = Philox counter of its next one (uint32 bit pattern; not a
        # reference quantity, see mg_quadrotor_autoreset)
        self.episode = torch.zeros(N, dtype=torch.int32, device=dev)
        self._state = _lib.QuadrotorState(*[_lib.ptr(t) for t in (self.pos, self.vel, self.omega, self.propw,
                                                                  self.rot, self.ct, self.episode)])
        # --- outputs ------------------------------------------------------------------------
        self.obs_dim = 19 if task == "velocity_control" else 16          # env.py:87-95
        self._obs = torch.zeros(N, self.obs_dim, dtype=torch.float32, device=dev)
        self._reward = torch.zeros(N, dtype=torch.float32, device=dev)
        # the reference returns a python float (f64); the exact copy costs 8 B/env of stores per step
        self._reward64 = torch.zeros(N, dtype=torch.float64, device=dev) if exact_reward else None
        self._done = torch.zeros(N, dtype=torch.bool, device=dev)    # kernel writes 0/1 bytes
        self._failed = torch.zeros(N, dtype=torch.uint8, device=dev)
        self._out_ptrs = [_lib.ptr(t) for t in (self._obs, self._reward, self._reward64, self._done, self._failed)]
        self._info_obj = _LazyInfo(self._obs, self._failed)

        # --- config -------------------------------------------------------------------------
        self._cfg = _lib.QuadrotorConfig()
        _fill_config(self._cfg, self.sim_config, dt, nt, task, healthy_reward)
        self.x_offset = self.y_offset = 0
        self.z_offset = 0.0
        self._map_t = None
        self.velocity_targets = None
        if task == "velocity_control":
            # env.py:99-102 -> QuadrotorSim.define_velocity_control_task(dt, nt, seed), quadrotorsim.py:306-319:
            # nt random-action steps from the pre-reset all-float32 zero state; same stream as
            # np.random.seed(seed) (a private RandomState: the reference reseeds numpy's global RNG here)
            rs = np.random.RandomState(seed)
            acts = np.stack([rs.uniform(low=self._cfg.min_voltage, high=self._cfg.max_voltage, size=4)
                             .astype(np.float32) for _ in range(int(nt))])
            a_t = torch.from_numpy(acts).to(dev).contiguous()
            self.velocity_targets = torch.zeros(int(nt), 3, dtype=torch.float32, device=dev)
            self._cfg.map_d, self._cfg.map_h, self._cfg.map_w = None, 0, 0
            rc = self._lib.mg_quadrotor_velocity_targets(self._cfg, int(nt), _lib.ptr(a_t),
                                                         _lib.ptr(self.velocity_targets), _lib.current_stream(dev))
            _lib.check(rc, "mg_quadrotor_velocity_targets")
            self._cfg.velocity_targets_d = self.velocity_targets.data_ptr()
            self.map_matrix = None
        else:
            self.map_matrix = Quadrotor.load_map(map_file)             # env.py:104
            ys, xs = np.where(self.map_matrix == -1)
            assert len(ys) == 1
            self.y_offset, self.x_offset = int(ys[0]), int(xs[0])
            self.z_offset = 5.0                                        # env.py:113
            self.map_matrix[self.y_offset, self.x_offset] = 0
            if map_file is not None or self.map_matrix.any():
                self._map_t = torch.from_numpy(self.map_matrix.astype(np.int32)).to(dev).contiguous()
                self._cfg.map_d = self._map_t.data_ptr()
            else:
                self._cfg.map_d = None                                 # flat floor fast path
            self._cfg.map_h, self._cfg.map_w = self.map_matrix.shape
        self._cfg.x_offset, self._cfg.y_offset, self._cfg.z_offset = self.x_offset, self.y_offset, self.z_offset

        self.valid_range = float(self.sim_config["fail"]["range"])
        lo, hi = self._cfg.min_voltage, self._cfg.max_voltage
        self.action_space = Box(low=np.array([lo] * 4, dtype="float32"),
                                high=np.array([hi] * 4, dtype="float32"), shape=[4])
        self.observation_space = Space(shape=[self.obs_dim], dtype="float32")
        # numpy's legacy seeding takes 32 bits; the fused auto-reset below uses all 64 of a larger seed
        self.np_random = np.random.RandomState(None if seed is None else int(seed) & 0xFFFFFFFF)
        self.seed_value = seed
        # fused auto-reset: done envs restart inside the step launch, noise from device-side Philox
        self.auto_reset = bool(auto_reset)
        self._ar = _lib.QuadrotorAutoReset()
        cv = self.sim_config.get("init_velocity")
        cw = self.sim_config.get("init_angular_velocity") if cv is not None else None
        for i, ax in enumerate("xyz"):
            self._ar.init_velocity[i] = float(cv[ax]) if cv else 0.0
            self._ar.init_angular_velocity[i] = float(cw[ax]) if cw else 0.0
        self._ar.init_velocity_noisy = float(cv["noisy"]) if cv else 0.0
        self._ar.init_angular_velocity_noisy = float(cw["noisy"]) if cw else 0.0
        self._ar.seed = int(seed or 0) & 0xFFFFFFFFFFFFFFFF
        self._ar.env_id_base = int(env_id_base)   # global id of env 0 when this batch is one shard of a larger job
        # cfg is final: fold it once (mg_quadrotor_plan_init); step()/rollout() then only enqueue the launch
        self._plan = _lib.QuadrotorPlan()
        with torch.cuda.device(dev):
            rc = self._lib.mg_quadrotor_plan_init(self._plan, self._cfg, self._ar if self.auto_reset else None, N,
                                                  self._state)
        _lib.check(rc, "mg_quadrotor_plan_init")
        self._plan_ref = C.byref(self._plan)
        self._plan_step = self._lib.mg_quadrotor_plan_step
        self._dev_index = dev.index if dev.index is not None else torch.cuda.current_device()
        self._action_shape = (N, 4)

    # ------------------------------------------------------------------ reference API
    def reset(self, mask=None, seed=None, init_velocity=None, init_angular_velocity=None):
        """Reset the envs selected by `mask` (bool/uint8 [N] tensor; None = all) and return obs [N,16].

        Noise (quadrotorsim.py:241-254): per env, four draws of random(3) in the reference's order
        (sign_v, mag_v, sign_w, mag_w) from `RandomState(seed)`; pass `init_velocity` /
        `init_angular_velocity` ([N,3] float64) to inject explicit values instead.
        """
        N, dev = self.num_envs, self.device
        if seed is not None:
            self.np_random = np.random.RandomState(seed)
        if init_velocity is None or init_angular_velocity is None:
            u = self.np_random.random_sample((N, 4, 3))
            cv, cw = self.sim_config.get("init_velocity"), self.sim_config.get("init_angular_velocity")
            v = np.zeros((N, 3))
            w = np.zeros((N, 3))
            if cv is not None:
                base = np.array([cv["x"], cv["y"], cv["z"]], dtype=np.float32)
                v = base + (float(cv["noisy"]) * u[:, 1]) * ((u[:, 0] > 0.5).astype(int) * 2 - 1.0)
            if cv is not None and cw is not None:    # the reference guards both blocks with init_velocity
                base = np.array([cw["x"], cw["y"], cw["z"]], dtype=np.float32)
                w = base + (float(cw["noisy"]) * u[:, 3]) * ((u[:, 2] > 0.5).astype(int) * 2 - 1.0)
            init_velocity = v if init_velocity is None else init_velocity
            init_angular_velocity = w if init_angular_velocity is None else init_angular_velocity
        iv = torch.as_tensor(np.ascontiguousarray(np.asarray(init_velocity, np.float64).T), device=dev)
        iw = torch.as_tensor(np.ascontiguousarray(np.asarray(init_angular_velocity, np.float64).T), device=dev)
        assert iv.shape == (3, N) and iw.shape == (3, N)
        m = None
        if mask is not None:
            m = torch.as_tensor(mask, device=dev).to(torch.uint8).contiguous()
            assert m.shape == (N,)
        rc = self._lib.mg_quadrotor_reset(self._cfg, N, self._state, _lib.ptr(m), _lib.ptr(iv), _lib.ptr(iw),
                                          _lib.ptr(self._obs), _lib.current_stream(dev))
        _lib.check(rc, "mg_quadrotor_reset")
        return self._obs

    def step(self, action):
        """action: float32 [N,4] tensor (or array-like). Returns (obs [N,16] f32, reward [N] f32,
        done [N] bool, info dict of [N] tensors). The four are views of persistent output buffers unless the env was built
        with `copy_outputs=True`: `.clone()` what you keep across steps."""
        a = action
        if not (isinstance(a, torch.Tensor) and a.dtype == torch.float32 and a.device == self.device
                and a.is_contiguous()):
            a = torch.as_tensor(action, dtype=torch.float32, device=self.device).contiguous()
        if a.shape != self._action_shape:
            raise ValueError("action must be [num_envs, 4], got %s" % (tuple(a.shape),))
        # one ctypes call per step and nothing else on the host: the constants were folded at construction
        # (mg_quadrotor_plan_init), the library selects the state's device itself, and no argument depends on
        # a step counter — so this launch can also be captured into a hipGraph as it stands
        # (host cost matters: at 65 536 envs the kernel is ~14 us, and a Python loop that issues launches more slowly than
        # that leaves the GPU idle between steps. The raw-stream query is the C-level form of
        # torch.cuda.current_stream(dev).cuda_stream without the Stream object, ~1.5 us less per step.)
        o = self._out_ptrs
        rc = self._plan_step(self._plan_ref, 1, a.data_ptr(), o[0], o[1], o[2], o[3], o[4], _raw_stream(self._dev_index))
        if rc != 0:
            _lib.check(rc, "mg_quadrotor_plan_step")
        if self.copy_outputs:
            obs = self._obs.clone()
            return obs, self._reward.clone(), self._done.clone(), _LazyInfo(obs, self._failed.clone())
        return self._obs, self._reward, self._done, self._info_obj

    def rollout(self, actions):
        """actions float32 [T,N,4]: T env-steps in one kernel launch (state stays in registers).
        Returns obs [T,N,16], reward [T,N], done [T,N] bool, failed [T,N] uint8."""
        a = torch.as_tensor(actions, dtype=torch.float32, device=self.device).contiguous()
        T, N = a.shape[0], self.num_envs
        assert a.shape == (T, N, 4)
        dev = self.device
        obs = torch.empty(T, N, self.obs_dim, dtype=torch.float32, device=dev)
        rew = torch.empty(T, N, dtype=torch.float32, device=dev)
        rew64 = torch.empty(T, N, dtype=torch.float64, device=dev)
        done = torch.empty(T, N, dtype=torch.bool, device=dev)
        failed = torch.empty(T, N, dtype=torch.uint8, device=dev)
        rc = self._plan_step(self._plan_ref, T, _lib.ptr(a), _lib.ptr(obs), _lib.ptr(rew), _lib.ptr(rew64),
                             _lib.ptr(done), _lib.ptr(failed), _lib.current_stream(dev))
        _lib.check(rc, "mg_quadrotor_plan_step")
        self._last_rollout_reward64 = rew64
        return obs, rew, done, failed

    def render(self, mode="human"):
        raise NotImplementedError("rendering is out of scope for the batched engine (SURVEY.md §2 row 5)")

    def close(self):
        pass

    # ------------------------------------------------------------------ extras
    @property
    def reward64(self):
        """float64 copy of the last step's reward (the reference returns a python float)."""
        return self._reward64

    _STATE_KEYS = ("pos", "vel", "omega", "propw", "rot", "ct", "episode")

    def state_dict(self):
        """Everything a bit-identical continuation needs: the simulator arrays, the per-env auto-reset counters
        (the Philox stream position of each env) and the host RandomState behind reset()."""
        sd = {k: getattr(self, k).clone() for k in self._STATE_KEYS}
        sd["np_random"] = self.np_random.get_state()
        return sd

    def load_state_dict(self, sd):
        for k in self._STATE_KEYS:
            if k == "episode" and k not in sd:
                continue                           # checkpoints of ABI 1 carried no counters
            dst = getattr(self, k)
            src = torch.as_tensor(sd[k])
            if tuple(src.shape) != tuple(dst.shape):
                raise ValueError("state_dict[%r] has shape %s, this env holds %s" % (k, tuple(src.shape), tuple(dst.shape)))
            dst.copy_(src.to(dst.dtype))
        if "np_random" in sd:
            self.np_random.set_state(sd["np_random"])

    @staticmethod
    def load_map(map_file):
        """env.py:293-304"""
        if map_file is None:
            flatten_map = np.zeros([100, 100], dtype=np.int32)
            flatten_map[50, 50] = -1
            return flatten_map
        rows = []
        with open(map_file, "r") as f:
            for line in f.readlines():
                rows.append([int(i) for i in line.split(" ")])
        return np.array(rows)
