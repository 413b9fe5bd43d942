"""`A1GymEnv` — the reference's `quadrupedal-v0` (metagym/quadrupedal/envs/gym_envs/a1_gym_env.py) for N robots, with
everything it computes in Python on the device, composed in the reference's order:

    policy action -> ETGWrapper (+ previous ETG output, MonitorEnv.py:261-273) -> LaikagoPoseOffsetGenerator ->
    [ActionFilter] -> 13 x (ApplyAction -> PHYSICS -> ReceiveObservation) -> info (pose, rot_mat, foot FK, energy) ->
    sensors -> observation[37] -> RewardShaping -> reward, done

The robot file is NOT part of this package (`a1/a1.urdf` ships with pybullet_data, not with the reference): pass
`urdf=<path>` and the env runs it on this package's articulated-body engine (`A1Physics`, a1_physics.py — the reference's world
parameters, dynamics parity with PyBullet unpinned), or pass `physics=`, any batched simulator with this protocol:

    physics.reset(mask)      -> q[N,12], qd[N,12], base_quat[N,4] (relative to the initial orientation), rpy_rate[N,3] (body
                                frame) after Minitaur.Reset (minitaur.py:398-444)
    physics.substep(torques) -> the same four after one 2 ms pybullet.stepSimulation() with those motor torques
                                (tensors are `[N, k]`; a simulator that keeps its state component-major hands back
                                `SoA(tensor[k, N])` instead and no copy is made — the layout is declared, never inferred)
    physics.world()          -> dict(base=[N,3] GetBasePosition, contact=[N,4] foot-ground flags, bad=[N] number of non-foot
                                contact points) at the end of the env step
    physics.fused_step(command[12][N], actuators) -> log[13][43][N]   (optional) the whole 13-sub-step loop in one launch with
                                the PD motor model evaluated inside it (A1Actuators.StepFused)
    physics.set_terrain(boxes, default_pose)   (optional) called once with the static boxes of `task`'s terrain
                                (terrain.task_terrain — what the reference creates in its Bullet world) and the reset pose
                                [0, 0, 0.28 + add_height] (locomotion_gym_env.py:337)

`task` selects the terrain exactly as in the reference (locomotion_gym_env.py:309-325): its `env_info` stretches drive the
slope / stair handling of the reward; the boxes and the reset height are handed to the physics.

The composition (which time the ETG sees, the hidden zero-action step inside reset(), sensors before the observation,
the reward against last step's base and feet) is pinned end to end against the unmodified `A1GymEnv` running on a scripted
Bullet client: tests/golden/a1_env.npz, tests/test_a1_env_gpu.py."""
import ctypes as C
import os

import numpy as np
import torch

from .. import _lib
from .a1_actuators import A1Actuators, MotorControlMode
from .a1_wrappers import EtgActionPath, Param_Dict, RewardShaping, SensorStack
from .terrain import TASK_NAMES, task_terrain, upstair_terrain
from ..spaces import Box


SENSOR_MODE = {"dis": 1, "motor": 1, "imu": 1, "contact": 1, "footpose": 0, "ETG": 0}      # a1_gym_env.py:13
TERRAIN_MODES = ("stair-fix", "stair-var", "downstair", "slope", "random", "special", "upstair-random", "downstair-random",   # locomotion_gym_env.py:25-27
                 "downslope-random", "upslope-random", "balance_beam", "cliff", "hurdle", "cave")
# tasks whose terrain LocomotionGymEnv builds at its FIRST reset (locomotion_gym_env.py:309-325), after — and therefore over — a
# `hardset` terrain asked for in that same call
_FIRST_RESET_TASKS = ("stairslope", "stairstair", "slopestair", "slopeslope", "gallop", "cave", "balancebeam", "highstair")
# LaikagoPoseOffsetGenerator's action space per `action_space` mode (simple_openloop.py:124-135)
DYNAMIC_KEYS = ("control_latency", "footfriction", "basemass", "baseinertia", "legmass", "leginertia", "motor_kp", "motor_kd", "gravity")
_ACTION_BOXES = {0: ([0.2, 0.7, 0.7] * 4, [-0.2, -0.7, -0.7] * 4),
                 1: ([0.1, 0.5, 0.4] * 4, [-0.1, -0.3, -0.6] * 4),
                 2: ([0.1, 0.5, 0.4, 0.1, 0.5, 0.4] + [0.1] * 6, [-0.1, -0.3, -0.6, -0.1, -0.3, -0.6] + [-0.1] * 6),
                 3: ([0.1, 0.7, 0.7, 0.1, 0.7, 0.7] + [0.1] * 6, [-0.1, -0.7, -0.7, -0.1, -0.7, -0.7] + [-0.1] * 6)}


def _env_info_copy(env_info):
    """info["env_info"]: a list of [x0, x1, env_vec[7]] (locomotion_gym_env.py:76) — copied so a checkpoint does not alias it."""
    return [[float(x0), float(x1), np.array(vec, dtype=np.float64, copy=True)] for x0, x1, vec in env_info]


class A1GymEnv(object):
    def __init__(self, num_envs, physics=None, device="cuda:0", ETG=0, ETG_T=0.5, ETG_H=20, ETG_path="", ETG_w=None, ETG_b=None,
                 act_mode="traj", task="plane", normal=0, action_space=0, reward_param=Param_Dict, reward_p=1.0, vel_d=0.6,
                 filter_=0, control_latency=0.002, motor_kp=None, motor_kd=None, env_info=None,
                 motor_control_mode=MotorControlMode.POSITION, sensor_mode=SENSOR_MODE, auto_reset=False, urdf=None,
                 urdf_options=None, vel_mode="max", random_param=None, dynamic_param=None, random_dynamic=False, seed=0,
                 force_source=None, action_limit=(0.75, 0.75, 0.75), render=False, on_rack=False, gait=0, step_y=0.05,
                 terrain_slots=1, terrain_max_boxes=96, x_noise_source=None, per_robot_dynamics=False, dynamics_source=None,
                 gravity_sign=1.0, sensor_noise_source=None, **kwargs):
        # ---- the rest of the reference's constructor signature (a1_gym_env.py:19-40; `gym.make('quadrupedal-v0')` registers
        #      action_limit / render / on_rack / random_dynamic / ETG / ETG_T / ETG_H / ETG_path / task / dynamic_param,
        #      quadrupedal/__init__.py:9-20) ---------------------------------------------------------------------------------
        #   action_limit  accepted; it has NO effect in the reference either: env_builder.py:113 hands the trajectory generator a
        #                 literal 0.75 ("#origin action_limit=action_limit") and LaikagoPoseOffsetGenerator derives its action space
        #                 from `action_space` alone (simple_openloop.py:124-135)
        #   step_y        accepted; no effect in the reference: ETG_model stores it in self.base_foot for task "balance"
        #                 (ETG_model.py:90-94) but act_clip reads the module-level `base_foot` (:122)
        #   render, on_rack, gait   viewers, the rack constraint and GaitGeneratorWrapperEnv are not built: refused by name
        #   **kwargs      swallowed like the reference's own **kwargs (kept in `ignored_kwargs`)
        if render:
            raise NotImplementedError("render=True: rendering is out of scope for the batched engine (SURVEY.md §2)")
        if on_rack:
            raise _lib.MetaGymHipError("on_rack=True (the robot hung from a fixed constraint, minitaur.py:411-413) is not built")
        if gait != 0:
            raise _lib.MetaGymHipError("gait=%r: GaitGeneratorWrapperEnv (env_builder.py:92-94) is not built; gait=0 is the reference's default" % (gait,))
        if task == "heightfield":
            # locomotion_gym_env.py:97-98,160-164: HeightField(2)._generate_field = GEOM_HEIGHTFIELD from
            # "heightmaps/wm_height_out.png" (envs/utilities/heightfield.py:35-37,89-104, mesh scale .05/.05/1.8, friction 5),
            # a pybullet_data asset that is in neither tree; the engine's terrain path takes boxes only. Accepting the name and
            # running on flat ground (what this class did until round 4) is a silent wrong answer: refused like gait / on_rack.
            raise NotImplementedError(
                "task='heightfield': the reference builds a PyBullet GEOM_HEIGHTFIELD from pybullet_data's "
                "heightmaps/wm_height_out.png (envs/utilities/heightfield.py:89-104); neither the asset nor a height-field narrow "
                "phase exists here — use one of the box courses (%s) or 'plane'" % ", ".join(sorted(TASK_NAMES)))
        self.action_limit, self.step_y, self.ignored_kwargs = tuple(action_limit), float(step_y), dict(kwargs)
        # ---- dynamics: handed in explicitly (`dynamic_param`, locomotion_gym_env.py:349-380) or redrawn per robot at every reset
        #      (`random_dynamic`, :381-405; a1_dynamics.py) ---------------------------------------------------------------------
        dyn = dict(dynamic_param or {})
        for key in dyn:
            if key not in DYNAMIC_KEYS:
                raise _lib.MetaGymHipError("dynamic_param[%r]: the reference reads %s (locomotion_gym_env.py:354-380)" % (key, ", ".join(DYNAMIC_KEYS)))
        self.random_dynamic = bool(random_dynamic)
        if "control_latency" in dyn and not self.random_dynamic:
            control_latency = 0.001 * float(dyn["control_latency"])                 # milliseconds (:354-355)
        if "motor_kp" in dyn and "motor_kd" in dyn and not self.random_dynamic:
            motor_kp, motor_kd = dyn["motor_kp"], dyn["motor_kd"]
        footfriction, basemass_ratio = float(dyn.get("footfriction", 1.0)), float(dyn.get("basemass", 1.0))      # :338-339,356-359
        gravity = dyn.get("gravity")
        # per-link ratios, a tilted gravity, per-reset dynamic_param and random_dynamic all go through per-robot model rows
        per_robot = self.random_dynamic or per_robot_dynamics or any(k in dyn for k in ("baseinertia", "legmass", "leginertia")) or \
            (gravity is not None and (float(gravity[0]) != 0.0 or float(gravity[1]) != 0.0))
        if per_robot and physics is not None and not hasattr(physics, "enable_per_robot_dynamics"):
            raise _lib.MetaGymHipError("random_dynamic / per-link dynamic_param change the SIMULATOR's masses, inertias, friction and gravity per "
                                       "robot: let A1GymEnv build A1Physics from urdf=, or give your physics the hooks of A1Physics "
                                       "(enable_per_robot_dynamics, write_body_tables, write_gravity, write_foot_friction)")
        if physics is None and urdf is not None:       # the robot file on this repo's own articulated-body engine
            from .a1_physics import A1Physics
            opts = dict(urdf_options or {})
            if not per_robot:
                opts.setdefault("foot_friction", footfriction)
                opts.setdefault("base_mass_ratio", basemass_ratio)
                if gravity is not None:
                    opts.setdefault("gravity", -float(gravity[2]))
            physics = A1Physics(num_envs, urdf=urdf, device=device, **opts)
        elif physics is not None and not per_robot and (("footfriction" in dyn) or ("basemass" in dyn) or gravity is not None):
            if not hasattr(physics, "set_dynamics"):
                raise _lib.MetaGymHipError("dynamic_param footfriction / basemass / gravity change the SIMULATOR: give your physics a "
                                           "set_dynamics(dict) method, or let A1GymEnv build A1Physics from urdf=")
            physics.set_dynamics({k: dyn[k] for k in ("footfriction", "basemass", "gravity") if k in dyn})
        if physics is None:
            raise _lib.MetaGymHipError(
                "quadrupedal-v0 needs the robot: pass urdf=<path to a1/a1.urdf> (the file ships with pybullet_data, not with the "
                "reference; it then runs on this package's articulated-body engine, metagym_amd.quadrupedal.A1Physics) or your own "
                "batched simulator as physics=<object> (protocol: metagym_amd/quadrupedal/a1_env.py). Everything around the "
                "physics (motor model, latency, ETG, sensors, reward) runs on the GPU either way.")
        self.num_envs, self.device, self.physics = int(num_envs), _lib.canonical_device(device), physics
        self.task = task
        self.add_height, task_env_info, self.terrain_boxes = task_terrain(task)
        self.env_info = task_env_info if env_info is None else env_info
        self.default_pose = [0.0, 0.0, 0.28 + self.add_height]                      # locomotion_gym_env.py:337
        # Terrains: one course for the whole batch (terrain_slots = 1, what the reference's single env has), or a TABLE of
        # `terrain_slots` courses with a course index per robot, so that reset(mask=, hardset=True, ...) can give part of the
        # batch a new course the way the reference's reset(hardset=True, ...) gives its one robot a new one per episode
        self.terrain_slots, self._terrain_max_boxes = int(terrain_slots), int(terrain_max_boxes)
        self._first_reset = True
        if self.terrain_slots > 1:
            if not hasattr(physics, "set_terrain_table"):
                raise _lib.MetaGymHipError("terrain_slots > 1 needs physics.set_terrain_table / write_course / terrain_id (A1Physics has them)")
            physics.set_terrain_table(self.terrain_slots, self._terrain_max_boxes)
            physics.write_course(0, self.terrain_boxes)
            self.terrain_id = physics.terrain_id                                    # int32 [N], shared with the engine and the reward
            self._add_height_t = torch.zeros(self.terrain_slots, dtype=torch.float64, device=self.device)
            self._add_height_t[0] = float(self.add_height)
            self._courses = {0: (self.add_height, self.env_info)}
            physics.set_reset_pose(self.default_pose)
        elif hasattr(physics, "set_terrain"):
            physics.set_terrain(self.terrain_boxes, self.default_pose)
        if ETG and ETG_w is None and len(ETG_path) > 1 and os.path.exists(ETG_path):   # MonitorEnv.py:241-244
            saved = np.load(ETG_path)
            ETG_w, ETG_b = saved["w"], saved["b"]
        kw = {} if motor_kp is None else dict(motor_kp=motor_kp, motor_kd=motor_kd)
        # env_builder.py:44-52: 13 sub-steps of 2 ms, no action interpolation, no command clip
        self.robot = A1Actuators(num_envs, device, time_step=0.002, action_repeat=13, control_latency=control_latency,
                                 motor_control_mode=motor_control_mode, enable_action_filter=bool(filter_), **kw)
        self.path = EtgActionPath(num_envs, device, ETG=ETG, ETG_T=ETG_T, ETG_H=ETG_H, ETG_w=ETG_w, ETG_b=ETG_b, act_mode=act_mode,
                                  task_mode="gallop" if task == "gallop" else "normal", action_space=action_space)
        self.sensors = SensorStack(num_envs, device, normal=normal, noise=bool(dict(sensor_mode).get("noise")), seed=seed,
                                   noise_source=sensor_noise_source)
        self.shaping = RewardShaping(num_envs, device, param=reward_param, reward_p=reward_p, vel_d=vel_d, env_info=self.env_info,
                                     vel_mode=vel_mode)
        if self.terrain_slots > 1:
            self._seg_table = torch.zeros(self.terrain_slots, _lib.A1_MAX_SEGMENTS, 5, dtype=torch.float64, device=self.device)
            self._seg_count = torch.zeros(self.terrain_slots, dtype=torch.int32, device=self.device)
            rows, cnt = RewardShaping.pack_env_info(self.env_info)
            self._seg_table[0].copy_(torch.as_tensor(rows))
            self._seg_count[0] = cnt
            self.shaping.set_terrain_table(self._seg_table, self._seg_count, self.terrain_id)
        hi, lo = _ACTION_BOXES[int(action_space)]
        self.action_space = Box(np.asarray(lo, np.float32), np.asarray(hi, np.float32), dtype=np.float32)   # simple_openloop.py:137
        # reset(yaw=, x_noise=): start heading and position noise per robot (locomotion_gym_env.py:327-338)
        self._x_noise = torch.zeros(self.num_envs, dtype=torch.bool, device=self.device)
        self._x_noise_any = False
        self._x_gen = torch.Generator(device=self.device)
        self._x_gen.manual_seed(int(seed) + 0xa11)
        self._x_noise_source = x_noise_source        # tests: a callable returning the add_x draws ([N] or scalar) instead of the generator
        self._yaw_init = torch.zeros(self.num_envs, dtype=torch.float64, device=self.device)
        self._pose_dirty = False
        self._lib = _lib.load()
        self.last_torques = None
        # the physics' one-launch path (fused_step) takes every motor mode and per-robot gains when it is A1Physics; a caller's
        # own fused_step is only assumed to know the reference's default (POSITION, shared gains) unless it says `fused_modes = "all"`
        self._fusable = (motor_control_mode is MotorControlMode.POSITION and motor_kp is None) or \
            getattr(physics, "fused_modes", None) == "all"
        # sub-steps since reset, on the device (exact in float64): the ETG's clock without a host value inside step(), so a
        # step can be captured into a hipGraph (capture_step)
        self._substeps_dev = torch.zeros(self.num_envs, dtype=torch.float64, device=self.device)
        # robots whose episode ended at the last step (auto_reset): one persistent buffer, so a captured step keeps reading it
        self.auto_reset = bool(auto_reset)
        self._pending = torch.zeros(self.num_envs, dtype=torch.bool, device=self.device)
        self._u8_one = torch.ones(self.num_envs, dtype=torch.uint8, device=self.device)
        self._u8_two = torch.full((self.num_envs,), 2, dtype=torch.uint8, device=self.device)
        # ---- RandomWrapper (MonitorEnv.py:521-662): pushes on the base, and what the observation reports about the dynamics ----
        f64 = dict(dtype=torch.float64, device=self.device)
        self._random_force = bool((random_param or {}).get("random_force"))
        # random_param["random_dynamics"]: accepted and, exactly like in the reference, without effect — RandomWrapper.random_dynamics
        # (MonitorEnv.py:547-629) is the only reader of the flag and both of its call sites are commented out (:632, :641)
        self._random_dynamics_flag = bool((random_param or {}).get("random_dynamics"))
        if self._random_force and not hasattr(physics, "apply_external_force"):
            raise _lib.MetaGymHipError("random_param['random_force'] needs physics.apply_external_force(force[N,3], position[N,3]) "
                                       "(pybullet.applyExternalForce on the base, LINK_FRAME; A1Physics has it)")
        self._force_pos, self._force_vec = torch.zeros(self.num_envs, 3, **f64), torch.zeros(self.num_envs, 3, **f64)
        self._force_on = torch.zeros(self.num_envs, dtype=torch.bool, device=self.device)      # a push precedes the next env step
        self._env_steps = torch.zeros(self.num_envs, dtype=torch.int64, device=self.device)    # LocomotionGymEnv._env_step_counter, per robot
        self._force_gen = torch.Generator(device=self.device)
        self._force_gen.manual_seed(int(seed) + 0x5eed)
        self._force_source = force_source            # tests: a callable returning the (position, force) draws instead of the generator
        self._force_scale = torch.tensor([0.2, 0.05, 0.05], **f64)
        self._force_dir_scale = torch.tensor([0.5, 1.0, 0.05], **f64)
        base_mass = getattr(physics, "base_mass", None)
        lat = torch.as_tensor(control_latency, **f64).expand(self.num_envs) if not torch.is_tensor(control_latency) else control_latency.to(**f64)
        self._dynamics = None if base_mass is None else torch.stack(                     # info["dynamics"] MonitorEnv.py:632: latency, foot friction, base mass
            [lat, torch.full((self.num_envs,), footfriction, **f64), torch.full((self.num_envs,), float(base_mass), **f64)], dim=1)
        # per-robot dynamics: every robot's own model row / gravity / foot friction in the engine, own latency and gains in the
        # actuators. random_dynamic: a fresh set per robot at each of ITS resets; else the constructor's dynamic_param, which a
        # reset(dynamic_param=) replaces for that one episode (locomotion_gym_env.py:349-352)
        self.dynamics = None
        if per_robot:
            from .a1_dynamics import A1Dynamics
            self.dynamics = A1Dynamics(physics, seed=seed, source=dynamics_source, gravity_sign=gravity_sign)
            kp0, kd0, _, _ = self.robot.motor_model_parameters()
            self._dyn_nominal = dict(control_latency=float(control_latency), motor_kp=np.asarray(kp0, np.float64), motor_kd=np.asarray(kd0, np.float64))
            self._dyn_default = self._fixed_dynamics(dyn)
            self._dyn_hold = torch.zeros(self.num_envs, dtype=torch.bool, device=self.device)      # robots carrying a reset(dynamic_param=) set
            self._dyn_override_seen = False
            self.dynamics.latency.fill_(float(control_latency))
            self.dynamics.motor_kp.copy_(torch.as_tensor(np.asarray(kp0, np.float64), **f64).expand(self.num_envs, 12))
            self.dynamics.motor_kd.copy_(torch.as_tensor(np.asarray(kd0, np.float64), **f64).expand(self.num_envs, 12))
            self.robot.SetControlLatency(self.dynamics.latency.clone())                              # per-robot tensors from here on
            self.robot.SetMotorGains(self.dynamics.motor_kp.clone(), self.dynamics.motor_kd.clone())
            self._fusable = getattr(physics, "fused_modes", None) == "all"             # per-robot gains inside the physics launch
            if not self.random_dynamic:
                self.dynamics.apply(self._dyn_default)
            self._push_dynamics()
        self._configure_observation(dict(sensor_mode), bool(ETG), int(ETG_H), int(normal))
        self.observation_space = Box(-np.inf * np.ones(self.observation_width, np.float32), np.inf * np.ones(self.observation_width, np.float32),
                                     dtype=np.float32)

    # ---- reset(**kwargs) of the reference: locomotion_gym_env.py:297-338 (terrain, yaw, x_noise), MonitorEnv.py:250-253 (ETG) ----
    RESET_KEYS = ("yaw", "x_noise", "ETG_w", "ETG_b", "hardset", "mode", "stepwidth", "slope", "stepheight", "env_vec", "info", "dynamic_param")

    def configure_reset(self, mask=None, terrain_rng=None, **kw):
        """What the reference's `reset(**kwargs)` changes BEFORE it resets the robot, for the robots in `mask` (device bool [N];
        None = all) — it takes effect at their next reset, explicit, masked or fused auto-reset:

          hardset=True, mode=, stepwidth=, slope=, stepheight=, env_vec=   a new terrain: the reference clears its Bullet world
                      (resetSimulation(), a new plane and a new robot object, locomotion_gym_env.py:238-276) and, for a `mode` of
                      its terrain_modes list, builds terrain.upstair_terrain(...) in it (:297-301) — add_height, env_info and the
                      boxes REPLACE these robots' course. (hardset=True without such a mode leaves the bare plane but keeps the old
                      add_height / env_info: replicated.) With terrain_slots = 1 only for the whole batch; else the course goes
                      into a free slot of the terrain table.
          yaw=        start heading (rad; scalar or [N]); x_noise=  truthy: every reset draws add_x = U(-0.2, 0.1) for the start
                      position (numpy's global stream in the reference; here a device generator, `x_noise_source` for tests)
          ETG_w=, ETG_b=   new ETG parameters (MonitorEnv.py:250-253) — one set for the whole batch.
          dynamic_param=   (envs with per-robot dynamics) this episode's dynamics for these robots instead of the constructor's
                      (locomotion_gym_env.py:349-352); installed right away — call it directly before the reset. Never read when
                      random_dynamic is set, like in the reference."""
        for k in kw:
            if k not in self.RESET_KEYS:
                raise TypeError("reset() got an unexpected keyword %r (the reference's: %s)" % (k, ", ".join(self.RESET_KEYS)))
        N, d = self.num_envs, self.device
        m = None if mask is None else torch.as_tensor(mask, device=d).bool()
        every = torch.ones(N, dtype=torch.bool, device=d) if m is None else m
        if kw.get("dynamic_param") and not self.random_dynamic:      # (`if not self.random:` — with random_dynamic the keyword is never read)
            if self.dynamics is None:
                raise _lib.MetaGymHipError("reset(dynamic_param=...) gives single robots their own dynamics: construct the env with "
                                           "per_robot_dynamics=True (or random_dynamic / per-link dynamic_param, which imply it)")
            for key in kw["dynamic_param"]:
                if key not in DYNAMIC_KEYS:
                    raise _lib.MetaGymHipError("dynamic_param[%r]: the reference reads %s" % (key, ", ".join(DYNAMIC_KEYS)))
            self.dynamics.apply(self._fixed_dynamics(kw["dynamic_param"]), every)
            self._dyn_hold.logical_or_(every)
            self._dyn_override_seen = True
            self._push_dynamics()
        if kw.get("ETG_w") is not None or kw.get("ETG_b") is not None:               # MonitorEnv.py:250-253
            w = self.path.etg_w() if kw.get("ETG_w") is None else kw["ETG_w"]
            b = self.path.etg_b() if kw.get("ETG_b") is None else kw["ETG_b"]
            self.path.set_etg_parameters(w, b)
        first_task_terrain = self._first_reset and self.task in _FIRST_RESET_TASKS
        if kw.get("hardset") and not first_task_terrain:
            if "mode" in kw and kw["mode"] in TERRAIN_MODES:
                args = dict(stepwidth=kw["stepwidth"], slope=kw["slope"], stepheight=kw["stepheight"], mode=kw["mode"], env_vecs=kw["env_vec"])
                if terrain_rng is not None:
                    args["rng"] = terrain_rng
                add_height, env_info, boxes = upstair_terrain(**args)
                self._set_course(m, add_height, env_info, boxes)
            else:       # the world is cleared, nothing is built in it; add_height and env_info are not touched (:297-301)
                if m is None or self.terrain_slots <= 1:
                    self._set_course(m, self.add_height, self.env_info, [])
                else:
                    raise _lib.MetaGymHipError("reset(hardset=True) without a terrain `mode` for part of the batch: give a mode of %r" % (TERRAIN_MODES,))
        if "yaw" in kw or "x_noise" in kw or m is None:
            yaw = torch.as_tensor(kw.get("yaw", 0.0), dtype=torch.float64, device=d).expand(N)
            if "yaw" in kw or m is None:
                self._yaw_init.copy_(torch.where(every, yaw, self._yaw_init))
            if "x_noise" in kw or m is None:
                self._x_noise.copy_(torch.where(every, torch.full_like(every, bool(kw.get("x_noise", False))), self._x_noise))
                self._x_noise_any = bool(kw.get("x_noise", False)) or (m is not None and self._x_noise_any)
        if float(torch.as_tensor(kw.get("yaw", 0.0)).abs().max()) != 0.0 or kw.get("x_noise"):
            if not hasattr(self.physics, "set_reset_pose"):
                raise _lib.MetaGymHipError("reset(yaw=, x_noise=) need physics.set_reset_pose(pose, mask, yaw) (A1Physics has it)")
            self._pose_dirty = True

    def _set_course(self, m, add_height, env_info, boxes):
        """A new course (add_height, env_info, boxes) for the robots in `m` (None: the whole batch)."""
        d = self.device
        if self.terrain_slots <= 1:
            if m is not None:
                raise _lib.MetaGymHipError("a new terrain for PART of the batch needs a terrain table: construct the env with terrain_slots > 1")
            self.add_height, self.env_info, self.terrain_boxes = add_height, env_info, boxes
            self.default_pose = [0.0, 0.0, 0.28 + add_height]
            if hasattr(self.physics, "set_terrain"):
                self.physics.set_terrain(boxes, self.default_pose)
            rows, cnt = RewardShaping.pack_env_info(env_info)
            c = self.shaping._cfg
            c.n_segments = cnt
            for i in range(cnt):
                c.seg[i][:] = list(rows[i])
            return
        if m is None:
            slot = 0
        else:       # a slot no robot outside `m` stands on (one host read: building a terrain is host work anyway)
            used = set(int(x) for x in torch.unique(self.terrain_id[~m]).tolist())
            free = [s for s in range(self.terrain_slots) if s not in used]
            if not free:
                raise _lib.MetaGymHipError("all %d terrain slots are in use by robots outside the mask: construct with more terrain_slots" % self.terrain_slots)
            slot = free[0]
        self.physics.write_course(slot, boxes)
        rows, cnt = RewardShaping.pack_env_info(env_info)
        self._seg_table[slot].copy_(torch.as_tensor(rows, dtype=torch.float64, device=d))
        self._seg_count[slot] = cnt
        self._add_height_t[slot] = float(add_height)
        self._courses[slot] = (add_height, env_info)
        if m is None:
            self.terrain_id.zero_()
            self.add_height, self.env_info = add_height, env_info
        else:
            self.terrain_id.copy_(torch.where(m, torch.full_like(self.terrain_id, slot), self.terrain_id))

    def _fixed_dynamics(self, param):
        """A `dynamic_param` dict (locomotion_gym_env.py:354-380) as a set for A1Dynamics.apply: absent keys = the nominal robot
        (:340-346; latency / gains: the constructor's)."""
        p, nom = dict(param or {}), self._dyn_nominal
        gains = "motor_kp" in p and "motor_kd" in p                                     # :375-378: only when both are given
        return self.dynamics.fixed(control_latency=0.001 * float(p["control_latency"]) if "control_latency" in p else nom["control_latency"],
                                   footfriction=p.get("footfriction", 1.0), basemass=p.get("basemass", 1.0),
                                   baseinertia=p.get("baseinertia", (1.0, 1.0, 1.0)), legmass=p.get("legmass", (1.0, 1.0, 1.0)),
                                   leginertia=p.get("leginertia", (1.0,) * 12), motor_kp=p["motor_kp"] if gains else nom["motor_kp"],
                                   motor_kd=p["motor_kd"] if gains else nom["motor_kd"], gravity=p.get("gravity", (0.0, 0.0, -10.0)))

    def _push_dynamics(self):
        """A1Dynamics' per-robot latency and gains into the actuators' device arrays, and the [latency, foot friction, base mass]
        rows the observation / info report (MonitorEnv.py:632, locomotion_gym_env.py:451-453). In place: a captured step reads them."""
        dy, keep = self.dynamics, self.robot._keep
        keep["control_latency"].copy_(dy.latency)
        keep["kp"].copy_(dy.motor_kp.t())
        keep["kd"].copy_(dy.motor_kd.t())
        if self._dynamics is not None:
            self._dynamics.copy_(torch.stack([dy.latency, dy.footfriction, dy.basemass], dim=1))

    def _dynamics_for_reset(self, m):
        """The dynamics part of LocomotionGymEnv.reset (:340-412) for the robots in `m` (None: all): random_dynamic — a fresh draw
        each; else back to the constructor's dynamic_param unless configure_reset(dynamic_param=) just gave them this episode's."""
        if self.dynamics is None:
            return
        every = torch.ones(self.num_envs, dtype=torch.bool, device=self.device) if m is None else m
        if self.random_dynamic:
            self.dynamics.apply(self.dynamics.draw(), every)
        elif self._dyn_override_seen:
            self.dynamics.apply(self._dyn_default, every & ~self._dyn_hold)
            self._dyn_hold.logical_and_(~every)
        else:
            return
        self._push_dynamics()

    def _place_for_reset(self, m):
        """Hand the physics the reset pose of the robots in `m` (None: all): [add_x, 0, 0.28 + add_height of the robot's course] and
        the start heading (locomotion_gym_env.py:331-338). Only when something differs from the construction-time pose."""
        if not (self._x_noise_any or self.terrain_slots > 1 or self._pose_dirty) or not hasattr(self.physics, "set_reset_pose"):
            return
        self._pose_dirty = True         # (once a pose other than the construction-time one was handed over, keep handing them over)
        N, d = self.num_envs, self.device
        x = torch.zeros(N, dtype=torch.float64, device=d)
        if self._x_noise_any:
            if self._x_noise_source is not None:
                draw = torch.as_tensor(self._x_noise_source(), dtype=torch.float64, device=d).expand(N)
            else:
                draw = -0.2 + 0.3 * torch.rand(N, generator=self._x_gen, dtype=torch.float64, device=d)     # np.random.uniform(-0.2, 0.1)
            x = torch.where(self._x_noise, draw, x)
        z = (0.28 + self._add_height_t[self.terrain_id.long()]) if self.terrain_slots > 1 else torch.full((N,), float(self.default_pose[2]), dtype=torch.float64, device=d)
        self.physics.set_reset_pose(torch.stack([x, torch.zeros_like(x), z], dim=0), mask=m, yaw=self._yaw_init)

    # FootPoseSensor's normalisation constants, robot_sensors.py:601-606
    FOOTPOSE_MEAN = [1.7454079e-01, -1.5465108e-01, -2.0661314e-01, 1.7080666e-01, 1.6490668e-01, -2.0865265e-01,
                     -1.9902834e-01, -1.2880404e-01, -2.3593837e-01, -2.0215839e-01, 1.3673349e-01, -2.3642859e-01]
    FOOTPOSE_STD = [3.9058894e-02, 2.4757426e-02, 4.2747084e-02, 4.1128017e-02, 2.7591322e-02, 4.3003809e-02,
                    4.3018311e-02, 2.8423777e-02, 4.7990609e-02, 4.6113804e-02, 2.8037265e-02, 4.9409315e-02]

    def _configure_observation(self, mode, etg, etg_h, normal):
        """env_builder.py:62-80 picks the sensors from `sensor_mode`: BaseDisplacementSensor (dis), IMUSensor with all six
        channels (imu 1) or the three rates (imu 2), MotorAngleAccSensor (motor 1) or MotorAngleSensor (motor 2),
        FootContactSensor (contact 1) or SimpleFootForceSensor (contact 2: needs `world()["foot_force"]` from the physics),
        FootPoseSensor (footpose); 0 switches a sensor off. ObservationWrapper (MonitorEnv.py:77-221) appends its own
        entries, force_vec / dynamic_vec included. `noise`: Gaussian draws inside every sensor (SensorStack: a device generator
        instead of numpy's global stream; `sensor_noise_source` injects them)."""
        self._sensor_noise = bool(mode.get("noise"))
        sel = (int(mode.get("dis", 0)), int(mode.get("imu", 0)), int(mode.get("motor", 0)), int(mode.get("contact", 0)),
               int(bool(mode.get("footpose", 0))))
        if sel[1] not in (0, 1, 2) or sel[2] not in (0, 1, 2) or sel[3] not in (0, 1, 2) or sel[0] not in (0, 1):
            raise ValueError("sensor_mode %r: dis 0/1, imu 0/1/2, motor 0/1/2, contact 0/1/2 (env_builder.py:62-80)" % (mode,))
        self._sensor_sel = sel
        self._sensor_width = 3 * sel[0] + (0, 6, 3)[sel[1]] + (0, 24, 12)[sel[2]] + (0, 4, 8)[sel[3]] + 12 * sel[4]
        f64 = dict(dtype=torch.float64, device=self.device)
        self._fp_mean, self._fp_std = torch.tensor(self.FOOTPOSE_MEAN, **f64), torch.tensor(self.FOOTPOSE_STD, **f64)
        self._obs_force, self._obs_dyn = bool(mode.get("force_vec")), bool(mode.get("dynamic_vec"))
        if self._obs_dyn and self._dynamics is None:
            raise _lib.MetaGymHipError("sensor_mode['dynamic_vec'] reports [control latency, foot friction, base mass] (MonitorEnv.py:632): "
                                       "the physics must expose `base_mass` (A1Physics does)")
        self._extras = ((_lib.A1_EXTRA_ETG if etg and mode.get("ETG") else 0) | (_lib.A1_EXTRA_ETG_OBS if etg and mode.get("ETG_obs") else 0)
                        | (_lib.A1_EXTRA_YAW if mode.get("yaw") else 0))
        self._etg_h, self._normal = etg_h, normal
        width = self._sensor_width + (12 if self._extras & _lib.A1_EXTRA_ETG else 0) \
            + (etg_h if self._extras & _lib.A1_EXTRA_ETG_OBS else 0) + (2 if self._extras & _lib.A1_EXTRA_YAW else 0) \
            + (6 if self._obs_force else 0) + (3 if self._obs_dyn else 0)
        rnn = mode.get("RNN")
        self._rnn = None
        if rnn and rnn["time_steps"] > 0:                                              # MonitorEnv.py:126-134
            assert rnn["mode"] in ("stack", "GRU")
            self._rnn = (int(rnn["time_steps"]), int(rnn["time_interval"]), rnn["mode"])
            self._obs_history = torch.zeros(self._rnn[0] * self._rnn[1], self.num_envs, width, dtype=torch.float64, device=self.device)
        self.observation_width = width

    def _wrap_observation(self, obs, pose, etg_obs, d_yaw, on_reset):
        """ObservationWrapper.reset (MonitorEnv.py:136-179) / step (:181-221) on the sensor observation `[N, 37]`."""
        N, d = self.num_envs, self.device
        extra_w = (12 if self._extras & _lib.A1_EXTRA_ETG else 0) + (self._etg_h if self._extras & _lib.A1_EXTRA_ETG_OBS else 0) \
            + (2 if self._extras & _lib.A1_EXTRA_YAW else 0)
        parts = [obs]
        if self._extras:
            extra = torch.empty(N, extra_w, dtype=torch.float64, device=d)
            p = pose.t().contiguous()
            eo = None if etg_obs is None else etg_obs.t().contiguous()
            dy = None if d_yaw is None else torch.as_tensor(d_yaw, dtype=torch.float64, device=d).expand(N).contiguous()
            with torch.cuda.device(d):
                rc = self._lib.mg_a1_observation_extras(N, self._extras, self._normal, self._etg_h, _lib.ptr(self.path.last_ETG_act),
                                                        _lib.ptr(eo), _lib.ptr(p), _lib.ptr(dy), _lib.ptr(extra), _lib.current_stream(d))
            _lib.check(rc, "mg_a1_observation_extras")
            k = extra_w - (2 if self._extras & _lib.A1_EXTRA_YAW else 0)     # the yaw pair comes last, after force_vec / dynamic_vec
            parts.append(extra[:, :k])
        if self._obs_force:                                                    # MonitorEnv.py:150-152,194-196
            on = self._force_on.reshape(-1, 1).to(torch.float64)
            parts.append(torch.cat([self._force_pos / self._force_scale, self._force_vec / 50.0], dim=1) * on)
        if self._obs_dyn:                                                      # :154-156,198-200
            parts.append(self._dynamics)
        if self._extras & _lib.A1_EXTRA_YAW:
            parts.append(extra[:, extra_w - 2:])
        if len(parts) > 1:
            obs = torch.cat(parts, dim=1)
        if self._rnn is not None:
            steps, interval, mode = self._rnn
            if on_reset:
                self._obs_history.zero_()
            frames = [self._obs_history[t * interval].clone() for t in range(steps)] + [obs]
            if not on_reset:
                self._obs_history[:-1] = self._obs_history[1:].clone()
            self._obs_history[-1] = obs
            obs = torch.stack(frames, dim=1)                                           # [N, steps + 1, width] ("GRU")
            if mode == "stack":
                obs = obs.reshape(N, -1)
        return obs

    def _select_sensors(self, obs37, info, world):
        """The configured sensors' blocks in sensor-NAME order (locomotion_gym_env.py:621-632): BaseDisplacement,
        FootContactSensor | FootForceSensor, FootPoseSensor, IMU, MotorAngle | MotorAngleAcc. `obs37` is the default stack's
        observation (mg_a1_observation: dis 3, contact 4, imu 6, motor 24); the alternatives are values `info` already holds."""
        dis, imu, motor, contact, footpose = self._sensor_sel
        if self._sensor_sel == (1, 1, 1, 1, 0):
            return obs37
        parts = []
        if dis:
            parts.append(obs37[:, 0:3])
        if contact == 1:
            parts.append(obs37[:, 3:7])
        elif contact == 2:          # SimpleFootForceSensor robot_sensors.py:546-548 = GetFootContactsForce('simple') a1.py:325-356
            if "foot_force" not in world:
                raise _lib.MetaGymHipError("sensor_mode['contact'] = 2 (SimpleFootForceSensor) needs world()['foot_force'] "
                                           "([N, 4] normal-force magnitudes in newtons) from the physics")
            parts.append(torch.cat([world["contact"], world["foot_force"] / 100.0], dim=1))
        if footpose:                # FootPoseSensor :607-611
            fp = info["footposition"]
            parts.append((fp - self._fp_mean) / self._fp_std if self._normal else fp)
        if imu == 1:
            parts.append(obs37[:, 7:13])
        elif imu == 2:              # IMUSensor(channels dR dP dY) is built without `normal` (env_builder.py:67): raw rates
            parts.append(info["drpy"] + self.sensors._noise[6:9].t() if self._sensor_noise else info["drpy"])      # (+ its drpy draws)
        if motor == 1:
            parts.append(obs37[:, 13:37])
        elif motor == 2:            # MotorAngleSensor :74-84; its noise is N(0, 5e-3): the angle slots (sigma 1e-2) at half scale
            parts.append(info["joint_angle"] + 0.5 * self.sensors._noise[9:21].t() if self._sensor_noise else info["joint_angle"])
        return torch.cat(parts, dim=1) if parts else obs37[:, 0:0]

    def get_time_since_reset(self):
        return self.robot.GetTimeSinceReset()

    def _info(self):
        N, d = self.num_envs, self.device
        f64 = dict(dtype=torch.float64, device=d)
        o = dict(pose=torch.empty(3, N, **f64), rot_mat=torch.empty(9, N, **f64), footposition=torch.empty(12, N, **f64),
                 joint_angle=torch.empty(12, N, **f64), drpy=torch.empty(3, N, **f64), energy=torch.empty(N, **f64))
        with torch.cuda.device(d):
            rc = self._lib.mg_a1_info(C.byref(self.robot._cfg), N, C.byref(self.robot._st), _lib.ptr(o["pose"]), _lib.ptr(o["rot_mat"]),
                                      _lib.ptr(o["footposition"]), _lib.ptr(o["joint_angle"]), _lib.ptr(o["drpy"]),
                                      _lib.ptr(o["energy"]), _lib.current_stream(d))
        _lib.check(rc, "mg_a1_info")
        return {k: (v if v.dim() == 1 else v.t()) for k, v in o.items()}

    # ---- RandomWrapper's pushes (MonitorEnv.py:530-535, 634-640, 644-660) --------------------------------------------------
    def _draw_force(self):
        """generate_randomforce() for every robot: position (U - 0.5) * 2 * (0.2, 0.05, 0.05), direction U(-1, 1)^3 * (0.5, 1, 0.05)
        normalised, magnitude U(20, 50). The reference draws from numpy's global stream; here a device generator seeded by the
        env (`seed`) — same distribution, not the same numbers."""
        if self._force_source is not None:
            pos, vec = self._force_source()
            f64 = dict(dtype=torch.float64, device=self.device)
            return torch.as_tensor(pos, **f64).expand(self.num_envs, 3), torch.as_tensor(vec, **f64).expand(self.num_envs, 3)
        u = torch.rand(self.num_envs, 7, generator=self._force_gen, dtype=torch.float64, device=self.device)
        pos = (u[:, 0:3] - 0.5) * 2.0 * self._force_scale
        v = (u[:, 3:6] * 2.0 - 1.0) * self._force_dir_scale
        return pos, v / v.norm(dim=1, keepdim=True) * (20.0 + 30.0 * u[:, 6:7])

    def _new_force(self, mask):
        """RandomWrapper.reset for the robots in `mask` (None: all): counter to 0, a fresh force, applied before the next env step."""
        if mask is None:
            self._env_steps.zero_()
        else:
            self._env_steps.masked_fill_(mask, 0)
        if not self._random_force:
            return
        pos, vec = self._draw_force()
        # (persistent buffers are updated IN PLACE: a captured step must find its state where it left it)
        if mask is None:
            self._force_pos.copy_(pos)
            self._force_vec.copy_(vec)
            self._force_on.fill_(True)
        else:
            m = mask.reshape(-1, 1)
            self._force_pos.copy_(torch.where(m, pos, self._force_pos))
            self._force_vec.copy_(torch.where(m, vec, self._force_vec))
            self._force_on.logical_or_(mask)

    def _force_after_step(self):
        """RandomWrapper.step after the inner env.step: the counter was incremented; a new force every 100 env steps, applied
        (again) while counter % 100 < 50. Nothing here reads a device value on the host."""
        self._env_steps += 1
        if not self._random_force:
            return
        c = self._env_steps % 100
        new, keep = c == 0, c < 50
        pos, vec = self._draw_force()
        self._force_pos.copy_(torch.where(new.reshape(-1, 1), pos, self._force_pos))
        self._force_vec.copy_(torch.where(new.reshape(-1, 1), vec, self._force_vec))
        self._force_on.copy_(new | keep)

    def _env_step(self, action, reset_mask=None, d_yaw=None, filter_init_mask=None):
        """LocomotionGymEnv.step below the wrappers (locomotion_gym_env.py:461-546)."""
        if self._random_force:      # applyExternalForce acts during the NEXT stepSimulation only: the first sub-step of this env step
            on = self._force_on.reshape(-1, 1).to(torch.float64)
            self.physics.apply_external_force(self._force_vec * on, self._force_pos * on)
        cmd, etg_obs = self.path.step(action, self._substeps_dev * self.robot.time_step)   # == get_time_since_reset()
        self._substeps_dev += 13.0
        if hasattr(self.physics, "fused_step") and self._fusable and self.robot.can_fuse():      # 13 sub-steps + PD model inside one physics launch
            self.last_torques = self.robot.StepFused(cmd, self.physics.fused_step, filter_init_mask=filter_init_mask)
        else:
            self.last_torques = self.robot.Step(cmd, self.physics.substep, filter_init_mask=filter_init_mask)
        self._force_after_step()
        world, info = self.physics.world(), self._info()
        info.update(base=world["base"], real_contact=world["contact"], bad=world["bad"], real_action=cmd, ETG_obs=etg_obs,
                    ETG_act=self.path.last_ETG_act.t())
        if self._random_force:
            info["force_vec"] = torch.cat([self._force_pos / self._force_scale, self._force_vec / 50.0], dim=1) \
                * self._force_on.reshape(-1, 1).to(torch.float64)
        if self._dynamics is not None:
            info["dynamics"] = self._dynamics
        obs = self.sensors.observe(world["base"], info["pose"], info["drpy"], info["joint_angle"], world["contact"], reset_mask)
        obs = self._select_sensors(obs, info, world)
        return self._wrap_observation(obs, info["pose"], etg_obs, d_yaw, False), info

    def reset(self, d_yaw=None, **kwargs):
        """A1GymEnv.reset(**kwargs) for every robot: the reference's keywords (`configure_reset`: hardset / mode / stepwidth / slope /
        stepheight / env_vec, yaw, x_noise, ETG_w, ETG_b — absent ones mean what they mean there: yaw 0, no x noise, terrain and
        ETG parameters unchanged), then robot.Reset and one observation, sensors reset, ETGWrapper.reset, then the hidden zero-action
        step of RewardShaping.reset (MonitorEnv.py:305-318) whose observation is the one returned. `d_yaw` only reaches
        ObservationWrapper.reset (the first frame of an RNN history): the hidden step runs without it. For part of the batch:
        `configure_reset(mask, ...)` and `step(action, reset_mask=mask)`."""
        N, d = self.num_envs, self.device
        self.configure_reset(None, **kwargs)
        self._first_reset = False
        self._place_for_reset(None)
        self._dynamics_for_reset(None)
        self.robot.Reset()
        self._pending.zero_()
        self._substeps_dev.zero_()
        self.robot.ReceiveObservation(*self.physics.reset(None))
        world, info = self.physics.world(), self._info()
        every = torch.ones(N, dtype=torch.bool, device=d)
        obs0 = self.sensors.observe(world["base"], info["pose"], info["drpy"], info["joint_angle"], world["contact"], every)
        obs0 = self._select_sensors(obs0, info, world)
        etg_obs0 = self.path.reset(self.get_time_since_reset())
        self._new_force(None)
        self._wrap_observation(obs0, info["pose"], etg_obs0, d_yaw, True)
        obs, _ = self._env_step(torch.zeros(N, 12, dtype=torch.float64, device=d))
        self.shaping.reset(world["base"], info["rot_mat"], info["footposition"])
        info.update(base=world["base"], real_contact=world["contact"], yaw_init=self._yaw_init.clone(), env_info=self.env_info)
        if self.terrain_slots > 1:
            info["terrain_id"] = self.terrain_id
        return obs, info

    def capture_step(self, warmup=2):
        """One `step(action)` as a hipGraph: the ~20 kernels of an env step (6 with a fused physics) are launch-bound at
        moderate batch sizes. Every launch of step() goes to torch's current stream through the C ABI and nothing in it reads a
        host value, so it captures as it stands. Returns `replay(action) -> (obs, reward, done, info)`; the outputs are the
        graph's static tensors (overwritten by the next replay). Capturing runs `warmup + 1` real steps: call reset()
        afterwards. Not for an env whose robot-level action filter still has to be initialised (first step after reset),
        nor with a yaw target."""
        d = self.device
        static_action = torch.zeros(self.num_envs, 12, dtype=torch.float64, device=d)
        side = torch.cuda.Stream(device=d)
        side.wait_stream(torch.cuda.current_stream(d))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self.step(static_action)
        torch.cuda.current_stream(d).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        if self._random_force and self._force_source is None:      # the pushes' generator advances inside the captured step
            graph.register_generator_state(self._force_gen)
        if self.sensors.noise and self.sensors._noise_source is None:
            graph.register_generator_state(self.sensors.noise_gen)
        if self.random_dynamic and self.auto_reset and self.dynamics.source is None:      # ... and so does the dynamics' one
            graph.register_generator_state(self.dynamics.gen)
        if self._x_noise_any and self._x_noise_source is None:      # reset(x_noise=True): add_x is drawn inside a reset the step performs
            graph.register_generator_state(self._x_gen)             # (an unregistered generator under capture raises or freezes)
        robot, repeat = self.robot, 13
        host_mirror = (robot._step_counter, robot._last_action)
        with torch.cuda.graph(graph):
            out = self.step(static_action)
        # capture RECORDS the launches without running them, but Step / StepFused advanced the host-side mirror of the device
        # clock (get_time_since_reset, the filter's `_step_counter == 0` test): put it back where the device is
        robot._step_counter, robot._last_action = host_mirror

        def replay(action):
            static_action.copy_(action)
            graph.replay()
            robot._step_counter += repeat          # the host-side mirror of the device clock
            return out
        replay.graph = graph
        return replay

    # ------------------------------------------------------------------ checkpoint
    def state_dict(self):
        """Everything a bit-identical continuation of the env needs, aggregated over its parts: the actuators (observation
        history ring, counters, last action), the robot-level action filter, the ETG's previous output, the sensor stack, the
        reward bookkeeping, the RNN frame history, the device-side sub-step clock, the pending auto-reset mask and — when the
        physics offers `state_dict()` (A1Physics does) — the simulator state."""
        sd = dict(robot=self.robot.state_dict(), etg_last_act=self.path.last_ETG_act.clone(),
                  sensors={k: t.clone() for k, t in self.sensors._t.items()}, shaping=self.shaping.state_dict(),
                  substeps=self._substeps_dev.clone(), pending=self._pending.clone(),
                  force=dict(pos=self._force_pos.clone(), vec=self._force_vec.clone(), on=self._force_on.clone(),
                             env_steps=self._env_steps.clone(), rng=self._force_gen.get_state()),
                  last_torques=None if self.last_torques is None else self.last_torques.clone())
        if self.sensors.noise:
            sd["sensor_noise_rng"] = self.sensors.noise_gen.get_state()
        if self.dynamics is not None:
            dy = self.dynamics
            sd["dynamics"] = dict(latency=dy.latency.clone(), footfriction=dy.footfriction.clone(), basemass=dy.basemass.clone(),
                                  motor_kp=dy.motor_kp.clone(), motor_kd=dy.motor_kd.clone(), rng=dy.gen.get_state(), hold=self._dyn_hold.clone(),
                                  override_seen=self._dyn_override_seen)
        if self.robot._action_filter is not None:
            sd["action_filter"] = dict(xhist=self.robot._action_filter.xhist.clone(), yhist=self.robot._action_filter.yhist.clone())
        if self._rnn is not None:
            sd["obs_history"] = self._obs_history.clone()
        if hasattr(self.physics, "state_dict"):
            sd["physics"] = self.physics.state_dict()
        # ---- what reset(**kwargs) / configure_reset() left behind (none of this used to be saved, so a checkpoint
        #      taken after hardset terrains, yaw, x_noise or new ETG parameters resumed on course 0 with the default pose) ----
        course = dict(add_height=self.add_height, env_info=_env_info_copy(self.env_info), default_pose=list(self.default_pose),
                      first_reset=self._first_reset)
        if self.terrain_slots > 1:       # (the engine's box table and terrain_id travel with the physics' state)
            course.update(seg_table=self._seg_table.clone(), seg_count=self._seg_count.clone(), add_height_t=self._add_height_t.clone(),
                          courses={k: (h, _env_info_copy(info)) for k, (h, info) in self._courses.items()},
                          terrain_id=self.terrain_id.clone())
        else:
            c = self.shaping._cfg
            course.update(segments=[[float(x) for x in c.seg[i]] for i in range(c.n_segments)],
                          boxes=[(list(map(float, h)), list(map(float, p)), list(map(float, q)), float(f)) for h, p, q, f in self.terrain_boxes])
        sd["course"] = course
        sd["start"] = dict(yaw=self._yaw_init.clone(), x_noise=self._x_noise.clone(), x_noise_any=self._x_noise_any,
                           pose_dirty=self._pose_dirty, x_rng=self._x_gen.get_state())
        sd["etg_parameters"] = dict(w=self.path.etg_w(), b=self.path.etg_b())
        return sd

    def load_state_dict(self, sd):
        self.robot.load_state_dict(sd["robot"])
        self.path.last_ETG_act.copy_(sd["etg_last_act"])
        for k, t in self.sensors._t.items():
            t.copy_(sd["sensors"][k])
        self.shaping.load_state_dict(sd["shaping"])
        self._substeps_dev.copy_(sd["substeps"])
        self._pending.copy_(sd["pending"])
        if "force" in sd:
            f = sd["force"]
            self._force_pos.copy_(f["pos"])
            self._force_vec.copy_(f["vec"])
            self._force_on.copy_(f["on"])
            self._env_steps.copy_(f["env_steps"])
            self._force_gen.set_state(f["rng"])
        self.last_torques = None if sd.get("last_torques") is None else sd["last_torques"].clone()
        if self.robot._action_filter is not None and "action_filter" in sd:
            self.robot._action_filter.xhist.copy_(sd["action_filter"]["xhist"])
            self.robot._action_filter.yhist.copy_(sd["action_filter"]["yhist"])
        if self._rnn is not None and "obs_history" in sd:
            self._obs_history.copy_(sd["obs_history"])
        if "physics" in sd and hasattr(self.physics, "load_state_dict"):
            self.physics.load_state_dict(sd["physics"])
        if "sensor_noise_rng" in sd and self.sensors.noise:
            self.sensors.noise_gen.set_state(sd["sensor_noise_rng"])
        if "dynamics" in sd and self.dynamics is not None:
            dy, f = self.dynamics, sd["dynamics"]
            for k in ("latency", "footfriction", "basemass", "motor_kp", "motor_kd"):
                getattr(dy, k).copy_(f[k])
            dy.gen.set_state(f["rng"])
            self._dyn_hold.copy_(f["hold"])
            self._dyn_override_seen = bool(f["override_seen"])
            self._push_dynamics()
        if "etg_parameters" in sd:
            self.path.set_etg_parameters(sd["etg_parameters"]["w"], sd["etg_parameters"]["b"])
        if "start" in sd:
            st = sd["start"]
            self._yaw_init.copy_(st["yaw"])
            self._x_noise.copy_(st["x_noise"])
            self._x_noise_any, self._pose_dirty = bool(st["x_noise_any"]), bool(st["pose_dirty"])
            self._x_gen.set_state(st["x_rng"])
        if "course" in sd:
            co = sd["course"]
            self.add_height, self.env_info = co["add_height"], _env_info_copy(co["env_info"])
            self.default_pose, self._first_reset = list(co["default_pose"]), bool(co["first_reset"])
            if self.terrain_slots > 1:
                if "seg_table" not in co:
                    raise ValueError("the checkpoint was taken with terrain_slots = 1, this env has a terrain table")
                self._seg_table.copy_(co["seg_table"])
                self._seg_count.copy_(co["seg_count"])
                self._add_height_t.copy_(co["add_height_t"])
                self._courses = {int(k): (h, _env_info_copy(info)) for k, (h, info) in co["courses"].items()}
                self.terrain_id.copy_(co["terrain_id"])       # (also restored by the physics when it is A1Physics: the same tensor)
            else:
                if "segments" not in co:
                    raise ValueError("the checkpoint was taken with a terrain table, this env has terrain_slots = 1")
                c = self.shaping._cfg
                c.n_segments = len(co["segments"])
                for i, row in enumerate(co["segments"]):
                    c.seg[i][:] = list(row)
                self.terrain_boxes = list(co["boxes"])
                # (the engine's boxes came back with the physics' state; a physics without state_dict gets them here)
                if hasattr(self.physics, "set_terrain") and not ("physics" in sd and hasattr(self.physics, "load_state_dict")):
                    self.physics.set_terrain(self.terrain_boxes, self.default_pose)

    def _begin_partial_reset(self, m):
        """The first half of A1GymEnv.reset() for the robots in `m` only (device bool `[N]`), everyone else untouched: robot
        reset and its one observation, sensor reset, ObservationWrapper.reset, ETGWrapper.reset at t = 0. Returns what
        RewardShaping.reset needs once the hidden zero-action step (= the env step this call opens) is done."""
        self._place_for_reset(m)
        self._dynamics_for_reset(m)
        self.robot.Reset(mask=m)
        self._substeps_dev.masked_fill_(m, 0.0)
        two = torch.where(m, self._u8_one, self._u8_two)                       # 1 reset, 2 untouched: one array for both kernels
        self.robot.ReceiveObservation(*self.physics.reset(m), only_mask=two)
        world, info = self.physics.world(), self._info()
        obs0 = self.sensors.observe(world["base"], info["pose"], info["drpy"], info["joint_angle"], world["contact"], two)
        obs0 = self._select_sensors(obs0, info, world)
        etg_obs0 = self.path.reset(self._substeps_dev * self.robot.time_step, mask=m)
        self._new_force(m)
        if self._rnn is not None:       # ObservationWrapper.reset for these robots: an empty frame history holding the reset observation
            rnn, self._rnn = self._rnn, None
            first = self._wrap_observation(obs0, info["pose"], etg_obs0, None, True)
            self._rnn = rnn
            mm = m.reshape(1, -1, 1)
            self._obs_history.copy_(torch.where(mm, torch.zeros_like(self._obs_history), self._obs_history))
            self._obs_history[-1] = torch.where(m.reshape(-1, 1), first, self._obs_history[-1])
        return world["base"], info["rot_mat"], info["footposition"]

    def step(self, action, d_yaw=None, reset_mask=None, donef=False, **kwargs):
        """A1GymEnv.step for every robot. `donef` is accepted like the reference's (MonitorEnv.py:343) and, like there, changes
        nothing: reward_shaping receives it and never reads it (:367-384). `reset_mask` (device bool `[N]`; with `auto_reset=True` the robots whose episode
        ended at the previous step): these robots run A1GymEnv.reset() INSIDE this step — robot / sensor / ETG reset, then the
        hidden zero-action env step of RewardShaping.reset (MonitorEnv.py:305-318), which is this very step: their action is
        ignored, the observation returned for them is the one reset() returns, reward 0, done False, info["reset"] True."""
        if kwargs:
            raise TypeError("step() got unexpected keywords %r (the reference's: d_yaw, donef)" % (sorted(kwargs),))
        m = reset_mask if reset_mask is not None else (self._pending if self.auto_reset else None)
        if m is not None:
            m = torch.as_tensor(m, device=self.device).bool()
            at_reset = self._begin_partial_reset(m)
            action = action.masked_fill(m.reshape(-1, 1), 0.0)
        obs, info = self._env_step(action, d_yaw=d_yaw, filter_init_mask=m)
        reward, done, terms = self.shaping.step(info["base"], info["pose"], info["rot_mat"], info["footposition"],
                                                info["real_contact"], info["energy"], info["bad"], d_yaw)
        info.update(terms)
        if m is not None:
            self.shaping.reset(*at_reset, mask=m)
            reward, done = reward.masked_fill(m, 0.0), done.masked_fill(m, False)
            info["reset"] = m.clone()       # (m may be the persistent auto-reset buffer, overwritten just below)
        if self.auto_reset:
            self._pending.copy_(done)
        return obs, reward, done, info
