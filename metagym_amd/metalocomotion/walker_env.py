"""Batched walker envs — host-side mirror of metagym/metalocomotion/envs:
`MetaHumanoidEnv` (meta_humanoids/meta_humanoids_env.py:7-40) and `MetaAntEnv`
(meta_ants/meta_ants_env.py:7-40) on top of `WalkerBaseEnv` (utils/walker_base_env.py).

Same surface: `sample_task(task_type)` returns a variant file name, `set_task(task_file)` loads it,
`reset()` / `step(action)` return the 44- (humanoid) or 28-dimensional (ant) observation, the summed
reward, done and `info = {"rewards": [N,5], "steps": [N]}`. Batched extensions: `set_task` also takes
a list of task files (or parsed `Model`s) plus `task_ids`, so env e can run its own body variant.

The reference ships 385 MJCF files per robot under metalocomotion/envs/assets/; this package does not copy
them, it regenerates the same 384 body variants from their generator patterns (`variants.py`), so
`sample_task('TRAIN' | 'TEST' | 'OOD')` returns the reference's file names and `set_task` accepts them on a
machine without the reference tree. `assets_dir` (or $METAGYM_LOCOMOTION_ASSETS) may still point at a directory
of MJCF files (the reference's, or your own robots of the same topology); `Model` objects
(metagym_amd.metalocomotion.load_mjcf) are accepted too.

Physics parity with the reference is UNPINNED (the reference calls PyBullet); see DESIGN.md §3.4. The envs default to
`preset="bullet"`, the closest known reading of that backend (bounding-box link inertia, inertial frame at the body origin,
0.04 / 0.04 body velocity damping, armature and stiffness ignored); `preset="mujoco"` is MuJoCo's documented reading of the
same files (what rounds 1-3 ran).
"""
import os
import random

import numpy as np
import torch

from .. import _lib
from ..spaces import Box
from . import variants
from .mjcf import DEFAULT_PRESET, Model, contact_margins, load_mjcf, preset_options


# The Bullet world parameters the reference sets (tests/test_walker_rules.py checks them against what its own scene code
# hands to the PyBullet stand-in): scene_bases.py:52-56 (gravity from env_bases.py:48, contact ERP, 4 sub-steps of 5 ms,
# 5 solver iterations), stadium.py:19-25 (ground lateral friction; Bullet multiplies it with the geom's).
CONTACT_ERP, GRAVITY, GROUND_FRICTION = 0.9, 9.8, 0.8


def pack_model(m, motor_torque, margins=None):
    """One row of the mg_walker_models table (layout documented in include/metagym_hip.h); `margins`: the per-proxy contact margins
    that ride behind the capsules when mg_walker_params.sphere_margin_in_table is set."""
    parts = [m.body_pos, m.body_rot, m.body_mass, m.body_com, m.body_inertia, m.joint_anchor, m.joint_axis,
             m.joint_lo, m.joint_hi, m.joint_armature, m.joint_damping, m.joint_stiffness, motor_torque,
             m.sph_pos, m.sph_radius, m.geom_p0, m.geom_p1, m.geom_radius]
    if margins is not None:
        parts.append(margins)
    return np.concatenate([np.asarray(p, np.float64).reshape(-1) for p in parts])


class WalkerBatchEnv(object):
    """Generic batched walker; subclasses fix the robot constants."""

    robot_dir = None
    variant_prefix = None
    foot_list = ()
    power = 1.0
    motor_power = None            # per joint, or None -> 100 (robot_bases.py:93 power_coef)
    alive_z, alive_bonus = 0.0, 1.0
    initial_z = None              # None -> height of the base at reset (walker_base.py:44-45)
    torque_f32 = False            # apply_action multiplies in float64 (walker_base.py:26-29) / float32 (humanoids.py:50-54)

    def __init__(self, num_envs=1, device="cuda", frame_skip=4, time_step=0.005, enable_render=False,
                 max_steps=2000, assets_dir=None, solver_iterations=5, mapping="wave", self_collision=True,
                 auto_reset=False, seed=0, env_id_base=0, gravity=GRAVITY, ground_friction=GROUND_FRICTION,
                 body_damping=None, per_proxy_friction=False, contact_erp=CONTACT_ERP, foot_force=False, preset=None,
                 max_coordinate_velocity=None, contact_margin=None):
        self._lib = _lib.load()
        self.num_envs = int(num_envs)
        self.device = _lib.canonical_device(device)
        if self.device.type != "cuda":
            raise _lib.MetaGymHipError("metagym_amd runs on an AMD GPU only (got device %r); there is no CPU "
                                       "path" % (device,))
        self.frame_skip, self.time_step, self.max_steps = int(frame_skip), float(time_step), int(max_steps)
        self.solver_iterations = int(solver_iterations)
        self.self_collision = bool(self_collision)     # the reference loads the MJCF with URDF_USE_SELF_COLLISION
        # World options beyond the MetaLocomotion defaults (all handled by the shape-generic wave kernels):
        #   gravity / ground_friction   the Bullet world's (quadrupedal: 10 and 5, locomotion_gym_env.py:251,258)
        #   preset                      'bullet' (default) | 'mujoco': how task files are read (mjcf.PRESETS: inertia rule,
        #                               inertial frame, armature, stiffness) and the default of `body_damping`
        #   body_damping = (lin, ang)   btMultiBody's velocity damping of every body; PyBullet's default is (0.04, 0.04), which
        #                               MetaLocomotion never changes — the 'bullet' preset's value; (0, 0) in the 'mujoco' one
        #   per_proxy_friction          every collision proxy carries its own link's coefficient (Model.sph_friction)
        self.gravity, self.ground_friction = float(gravity), float(ground_friction)
        self.preset = DEFAULT_PRESET if preset is None else preset
        if body_damping is None:
            body_damping = preset_options(self.preset)["body_damping"]
        self.body_damping = (float(body_damping[0]), float(body_damping[1]))
        #   max_coordinate_velocity     btMultiBody's clamp of every generalized velocity at the end of a sub-step (Bullet: 100,
        #                               the 'bullet' preset's value; 0 = off, the 'mujoco' one)
        if max_coordinate_velocity is None:
            max_coordinate_velocity = preset_options(self.preset)["max_velocity"]
        self.max_coordinate_velocity = float(max_coordinate_velocity)
        #   contact_margin              Bullet's contact-breaking margin (mg_walker_params.contact_margin / sphere_margin): a proxy
        #                               within it above the ground / a terrain box is a contact point — a speculative solver
        #                               row and a feet_contact flag (walker_base_env.py:57-63 reads getContactPoints).
        #                               "relative" = Bullet's default rule, 0.02 x the link's angular motion disc
        #                               (mjcf.contact_margins; 3 - 8 mm on the humanoid) — the 'bullet' preset; a float = that many
        #                               metres for every proxy; 0 (penetration only) in the 'mujoco' preset
        if contact_margin is None:
            contact_margin = preset_options(self.preset)["contact_margin"]
        self.contact_margin = contact_margin if isinstance(contact_margin, str) else float(contact_margin)
        self.per_proxy_friction = bool(per_proxy_friction)
        #   contact_erp                 0.9 for MetaLocomotion (scene_bases.py:55 setDefaultContactERP); a world that never calls it
        #                               keeps Bullet's default 0.2 (btContactSolverInfo::m_erp2) — the quadrupedal one
        self.contact_erp = float(contact_erp)
        #   foot_force                  also report each foot's normal-force magnitude (mg_walker_state.foot_force; quadrupedal
        #                               SimpleFootForceSensor, a1.py:325-356)
        self.want_foot_force = bool(foot_force)
        assert mapping in ("wave", "lane")
        self.mapping = mapping    # 'wave': one wavefront per env, LDS-resident (default); 'lane': one lane per env
        self.assets_dir = assets_dir or os.environ.get("METAGYM_LOCOMOTION_ASSETS")
        self.tra_tasks, self.tst_tasks, self.ood_tasks = [], [], []
        d = self._robot_assets()
        if self.variant_prefix is None:              # a robot without MetaLocomotion variant files (set_task takes Models)
            pass
        elif d and os.path.isdir(d):
            for f in sorted(os.listdir(d)):          # meta_humanoids_env.py:18-27
                if f.find(self.variant_prefix + "_var_tra") == 0:
                    self.tra_tasks.append(f)
                if f.find(self.variant_prefix + "_var_tst") == 0:
                    self.tst_tasks.append(f)
                if f.find(self.variant_prefix + "_var_ood") == 0:
                    self.ood_tasks.append(f)
        else:                                        # no asset directory: the same names, regenerated (variants.py)
            self.tra_tasks = variants.task_names(self.variant_prefix, "TRAIN")
            self.tst_tasks = variants.task_names(self.variant_prefix, "TEST")
            self.ood_tasks = variants.task_names(self.variant_prefix, "OOD")
        self._robot_set = False
        self.np_random = np.random.RandomState()
        # fused auto-reset: finished envs restart inside the step launch with device-side (Philox) joint
        # noise keyed by (seed, env_id_base + env, step), so a sharded job reproduces the unsharded one
        self.auto_reset, self.seed_value, self.env_id_base = bool(auto_reset), int(seed), int(env_id_base)
        self.global_step = 0

    def _robot_assets(self):
        return os.path.join(self.assets_dir, self.robot_dir) if self.assets_dir else None

    # ------------------------------------------------------------------ tasks
    def sample_task(self, task_type=None):
        """meta_humanoids_env.py:32-40"""
        if task_type is None or task_type == "TRAIN":
            return random.choice(self.tra_tasks)
        elif task_type == "TEST":
            return random.choice(self.tst_tasks)
        elif task_type == "OOD":
            return random.choice(self.ood_tasks)
        raise Exception("Unexpected task_type: %s" % task_type)

    def _to_model(self, t):
        if isinstance(t, Model):
            return t
        path = t if os.path.isabs(t) or self._robot_assets() is None else os.path.join(self._robot_assets(), t)
        if not os.path.exists(path):
            m = variants.model_from_task_name(t, preset=self.preset)     # one of the reference's file names: regenerate that variant
            if m is not None:
                return m
        return load_mjcf(path, foot_names=self.foot_list, preset=self.preset)

    def set_task(self, task_file, task_ids=None):
        tasks = task_file if isinstance(task_file, (list, tuple)) else [task_file]
        models = [self._to_model(t) for t in tasks]
        m0 = models[0]
        nb, nj, ns, nf = len(m0.body_parent), len(m0.joint_body), len(m0.sph_body), len(m0.foot_body)
        for m in models:
            assert (np.array_equal(m.body_parent, m0.body_parent) and np.array_equal(m.joint_body, m0.joint_body)
                    and np.array_equal(m.sph_body, m0.sph_body) and np.array_equal(m.geom_body, m0.geom_body)
                    and np.array_equal(m.pair_a, m0.pair_a)), "all tasks of a batch must share one topology"
        tp = _lib.WalkerTopology()
        tp.n_bodies, tp.n_joints, tp.n_spheres, tp.n_feet = nb, nj, ns, nf
        for i, v in enumerate(m0.body_parent): tp.body_parent[i] = int(v)
        for i, v in enumerate(m0.joint_body): tp.joint_body[i] = int(v)
        for i, v in enumerate(m0.sph_body): tp.sphere_body[i] = int(v)
        for i, v in enumerate(m0.foot_body): tp.foot_body[i] = int(v)
        # which feet_contact flag a proxy reports to: its own link's when the model says so (URDF robots with merged fixed
        # links), else the MetaLocomotion rule — a foot is a whole body (walker_base_env.py:57-63)
        sph_foot = getattr(m0, "sph_foot", None)
        if sph_foot is None:
            sph_foot = [next((f for f, fb in enumerate(m0.foot_body) if int(fb) == int(b)), -1) for b in m0.sph_body]
        for m in models:
            assert np.array_equal(getattr(m, "sph_foot", sph_foot), sph_foot), "all tasks of a batch must share one topology"
        for i, v in enumerate(sph_foot): tp.sphere_foot[i] = int(v)
        tp.n_geoms, tp.n_pairs = len(m0.geom_body), len(m0.pair_a)
        assert tp.n_geoms <= _lib.WALKER_MAX_GEOMS and tp.n_pairs <= _lib.WALKER_MAX_PAIRS
        assert nb <= _lib.WALKER_MAX_BODIES and nj <= _lib.WALKER_MAX_JOINTS and ns <= _lib.WALKER_MAX_SPHERES and nf <= _lib.WALKER_MAX_FEET
        for i, v in enumerate(m0.geom_body): tp.geom_body[i] = int(v)
        for i, (a, b) in enumerate(zip(m0.pair_a, m0.pair_b)): tp.pair_a[i], tp.pair_b[i] = int(a), int(b)
        self._topo = tp
        mp = np.full(nj, 100.0) if self.motor_power is None else np.asarray(self.motor_power, float)
        assert len(mp) == nj
        torque = mp * self.power                                   # humanoids.py:50-54, walker_base.py:26-29
        # Bullet's relative contact margin rule: every row carries its model's per-proxy margins
        rel = isinstance(self.contact_margin, str)
        table = np.stack([pack_model(m, torque, contact_margins(m, self.contact_margin) if rel else None) for m in models])
        N, dev = self.num_envs, self.device
        self._table = torch.from_numpy(table).to(dev).contiguous()
        ms = _lib.WalkerModels()
        ms.table, ms.n_tasks, ms.model_stride = self._table.data_ptr(), len(models), table.shape[1]
        self._models_c = ms
        self.models = models
        self.n_joints, self.n_feet = nj, nf
        self.obs_dim = 8 + 2 * nj + nf
        if task_ids is None:
            task_ids = torch.arange(N, dtype=torch.int32) % len(models)
        self.task_id = torch.as_tensor(task_ids, dtype=torch.int32).to(dev).contiguous()
        f64 = lambda *s: torch.zeros(*s, dtype=torch.float64, device=dev)
        self.pos, self.rot, self.vel, self.omega = f64(3, N), f64(9, N), f64(3, N), f64(3, N)
        self.q, self.qd, self.potential = f64(nj, N), f64(nj, N), f64(N)
        self.feet_contact = torch.zeros(nf, N, dtype=torch.float32, device=dev)
        self.steps = torch.zeros(N, dtype=torch.int32, device=dev)
        self.bad_contacts = torch.zeros(N, dtype=torch.int32, device=dev)    # contact points on proxies that are no foot
        st = _lib.WalkerState()
        for k in ("task_id", "pos", "rot", "vel", "omega", "q", "qd", "potential", "feet_contact", "steps", "bad_contacts"):
            setattr(st, k, getattr(self, k).data_ptr())
        self.ext_wrench = None      # optional [6][N] push on the base body, first sub-step of every launch (set_external_wrench)
        self.foot_force = torch.zeros(nf, N, dtype=torch.float64, device=dev) if self.want_foot_force else None
        st.foot_force = self.foot_force.data_ptr() if self.want_foot_force else None
        self._state_c = st
        p = _lib.WalkerParams()
        p.time_step, p.frame_skip, p.solver_iterations = self.time_step, self.frame_skip, self.solver_iterations
        # Bullet multiplies the two bodies' lateral friction: ground 0.8 (stadium.py:23) x geom friction (MJCF)
        p.limit_erp = 0.2                                          # Bullet's default constraint ERP
        p.erp, p.gravity = self.contact_erp, self.gravity
        self._geom_friction = float(m0.geom_friction)
        if self.per_proxy_friction:          # the plane's own coefficient; the kernel multiplies with the proxy's link
            mu = np.asarray(getattr(m0, "sph_friction", np.full(ns, float(m0.geom_friction))), np.float64)
            assert mu.shape == (ns,)
            self._sphere_friction = torch.from_numpy(mu.copy()).to(dev)
            p.friction, p.sphere_friction = self.ground_friction, self._sphere_friction.data_ptr()
        else:
            self._sphere_friction = None
            p.friction, p.sphere_friction = self.ground_friction * float(m0.geom_friction), None
        p.body_linear_damping, p.body_angular_damping = self.body_damping
        p.max_coordinate_velocity = self.max_coordinate_velocity
        p.contact_margin, p.sphere_margin_in_table = (0.0, 1) if rel else (self.contact_margin, 0)
        p.alive_z, p.alive_bonus, p.dead_bonus = self.alive_z, self.alive_bonus, -1.0
        p.initial_z = float(self.initial_z if self.initial_z is not None else m0.body_pos[0][2])
        p.joints_at_limit_cost = -0.1                              # walker_base_env.py:22
        p.walk_target_x, p.walk_target_y = 1e3, 0.0                # walker_base.py:9-10
        p.max_steps, p.floor_in_parts = self.max_steps, 1
        p.mapping = 1 if self.mapping == "wave" else 0
        p.self_collision = int(self.self_collision)
        p.self_friction = float(m0.geom_friction) ** 2            # Bullet multiplies the two geoms' friction
        p.auto_reset = int(self.auto_reset)
        p.seed, p.env_id_base = self.seed_value & 0xFFFFFFFFFFFFFFFF, self.env_id_base
        p.torque_f32 = int(self.torque_f32)
        p.height_f32 = int(self.initial_z is not None)     # python-float initial_z: float32 alive sum (walker_base_env.py:47)
        # the floor link joins robot.parts after an env's FIRST reset (walker_base_env.py:30-31) — per env: a masked first
        # reset must not change what the other envs' own first reset will see
        self._floor_known = torch.zeros(N, dtype=torch.bool, device=dev)
        self._all_floor_known = False                      # host mirror: True once every env went through a reset
        self._floor_known_host = np.zeros(N, dtype=bool)   # ... fed by masks that arrive on the host
        self._masked_first_resets = 0
        self._params_c = p
        self._apply_terrain()
        self._obs = torch.zeros(N, self.obs_dim, dtype=torch.float32, device=dev)
        self._reward = torch.zeros(N, dtype=torch.float32, device=dev)
        self._rewards5 = torch.zeros(N, 5, dtype=torch.float32, device=dev)
        self._done = torch.zeros(N, dtype=torch.bool, device=dev)
        high = np.ones([nj])
        self.action_space = Box(-high, high, dtype=np.float32)
        self.observation_space = Box(-np.inf * np.ones([self.obs_dim]), np.inf * np.ones([self.obs_dim]),
                                     dtype=np.float32)
        self._robot_set = True

    _STATE_KEYS = ("pos", "rot", "vel", "omega", "q", "qd", "potential", "feet_contact", "steps", "bad_contacts")

    def state_dict(self):
        """Simulator arrays + what a bit-identical continuation also needs: which model every env runs (task_id), the
        auto-reset stream position (global_step is the Philox step index), the host RandomState behind reset() and which envs
        have been reset at least once (the floor-in-parts quirk of the first reset, walker_base_env.py:30-31)."""
        sd = {k: getattr(self, k).clone() for k in self._STATE_KEYS}
        sd["task_id"] = self.task_id.clone()
        sd["floor_known"] = self._floor_known.clone()      # which envs' robot.parts already hold the floor link
        sd["global_step"] = int(self.global_step)
        sd["np_random"] = self.np_random.get_state()
        # static terrain: the shared box list, or the per-robot course table with every robot's course index (a
        # checkpoint taken on a terrain table used to resume every robot on course 0 of an empty table)
        tab = getattr(self, "_terrain_t", None)
        if tab is not None and tab.dim() == 3:
            sd["terrain_table"], sd["terrain_id"] = tab.clone(), self.terrain_id.clone()
        elif tab is not None:
            sd["terrain_boxes"] = tab.clone()
        # an explicit marker, so that loading a checkpoint taken WITHOUT terrain removes a terrain this env was built with
        sd["terrain_kind"] = "none" if tab is None else ("table" if tab.dim() == 3 else "boxes")
        return sd

    def load_state_dict(self, sd):
        for k in self._STATE_KEYS + (("task_id",) if "task_id" in sd else ()):
            if k == "bad_contacts" and k not in sd:
                continue                           # checkpoints written before ABI 4
            dst, src = getattr(self, k), torch.as_tensor(sd[k])
            if tuple(src.shape) != tuple(dst.shape):
                raise ValueError("state_dict[%r] has shape %s, this env holds %s" % (k, tuple(src.shape), tuple(dst.shape)))
            dst.copy_(src.to(dst.dtype))
        if "floor_known" in sd:
            self._floor_known.copy_(torch.as_tensor(sd["floor_known"]).to(self.device))
            self._all_floor_known = bool(self._floor_known.all())     # (load time, not on the step path)
            # the host bitmap of host-masked resets follows the checkpoint too: a stale all-True one would switch the remaining
            # envs' FIRST reset to floor_in_parts = 1
            self._floor_known_host = self._floor_known.cpu().numpy().astype(bool).copy()
            self._masked_first_resets = 0
        if "terrain_table" in sd:
            src = torch.as_tensor(sd["terrain_table"])
            tab = getattr(self, "_terrain_t", None)
            if tab is None or tab.dim() != 3 or tuple(tab.shape) != tuple(src.shape):
                self.set_terrain_table(src.shape[0], src.shape[1], terrain_id=getattr(self, "terrain_id", None))
            self._terrain_t.copy_(src.to(self.device))
            self.terrain_id.copy_(torch.as_tensor(sd["terrain_id"]).to(self.device))
        elif "terrain_boxes" in sd:
            # the checkpoint's rows BECOME the terrain spec: a later set_task() -> _apply_terrain() re-installs them
            # instead of the constructor's boxes (same-shape case) or nothing (different-shape case)
            self._terrain_spec, self._terrain_t = torch.as_tensor(sd["terrain_boxes"]).to(self.device).clone(), None
            self._apply_terrain()
        elif sd.get("terrain_kind") == "none":      # taken without terrain: whatever this env holds goes
            self._terrain_spec, self._terrain_t = None, None
            self._apply_terrain()
        if "global_step" in sd:
            self.global_step = int(sd["global_step"])
        if "np_random" in sd:
            self.np_random.set_state(sd["np_random"])

    # ------------------------------------------------------------------ episode control
    def reset(self, mask=None, seed=None, joint_noise=None):
        if not self._robot_set:
            raise Exception("BaseBulletEnv::_reset: must call set_robot and set_scene first")   # env_bases.py:68-69
        N, dev = self.num_envs, self.device
        if seed is not None:
            self.np_random = np.random.RandomState(seed)
        if joint_noise is None:                                    # walker_base.py:15, per env in joint order
            joint_noise = self.np_random.uniform(low=-0.1, high=0.1, size=(N, self.n_joints))
        if isinstance(joint_noise, _lib.JointMajor):               # already [n_joints][N] float64 on the device: taken as it is
            jn = joint_noise.t
        elif isinstance(joint_noise, torch.Tensor):                # a device tensor stays on the device (no host trip: capturable)
            jn = joint_noise.to(device=dev, dtype=torch.float64).t().contiguous()
        else:
            jn = torch.as_tensor(np.ascontiguousarray(np.asarray(joint_noise, np.float64).T), device=dev)
        assert jn.shape == (self.n_joints, N)
        m = None
        if mask is not None:
            if not (isinstance(mask, torch.Tensor) and mask.is_cuda) and not self._all_floor_known:
                # a mask that arrives on the host also updates the host bitmap of envs that went through a reset, so a user who
                # only ever resets with masks leaves the two-launch path as soon as every env had its first one
                self._floor_known_host |= np.asarray(mask.cpu() if isinstance(mask, torch.Tensor) else mask).astype(bool).reshape(N)
            m = torch.as_tensor(mask, device=dev)
            # (a contiguous bool tensor IS a 0 / 1 byte array: reinterpret, no kernel)
            m = m.view(torch.uint8) if (m.dtype == torch.bool and m.is_contiguous()) else m.to(torch.uint8).contiguous()
        # WalkerBaseEnv.reset adds the floor to robot.parts AFTER robot.reset() computed the reset observation and
        # potential (walker_base_env.py:24-31): the first reset after set_task averages over the robot's parts only
        # (per env: `_floor_known`. While some env has not been reset yet, the masked reset runs as two masked launches —
        # first-timers with floor_in_parts = 0, the others with 1 — so nothing here reads a device value on the host.)
        def launch(mask_t, floor):
            self._params_c.floor_in_parts = floor
            rc = self._lib.mg_walker_reset(self._topo, self._models_c, self._params_c, N, self._state_c, _lib.ptr(mask_t),
                                           _lib.ptr(jn), _lib.ptr(self._obs), _lib.current_stream(dev))
            self._params_c.floor_in_parts = 1
            _lib.check(rc, "mg_walker_reset")
        if self._all_floor_known:
            launch(m, 1)
        else:
            sel = torch.ones(N, dtype=torch.bool, device=dev) if m is None else m.bool()
            launch((sel & ~self._floor_known).to(torch.uint8).contiguous(), 0)
            launch((sel & self._floor_known).to(torch.uint8).contiguous(), 1)
            self._floor_known |= sel
            self._masked_first_resets += 1
            if m is None or bool(self._floor_known_host.all()):
                self._all_floor_known = True
            elif self._masked_first_resets % 64 == 0 and not torch.cuda.is_current_stream_capturing():
                # device masks: look at the device bitmap once in a while (one host read per 64 masked resets, never in a capture)
                self._all_floor_known = bool(self._floor_known.all())
        return self._obs

    def step(self, action):
        a = torch.as_tensor(action, dtype=torch.float32, device=self.device).contiguous()
        assert a.shape == (self.num_envs, self.n_joints), "action must be [num_envs, n_joints]"
        self._params_c.step_index = self.global_step
        self.global_step += 1
        rc = self._lib.mg_walker_step(self._topo, self._models_c, self._params_c, self.num_envs, self._state_c,
                                      _lib.ptr(a), _lib.ptr(self._obs), _lib.ptr(self._reward),
                                      _lib.ptr(self._rewards5), _lib.ptr(self._done),
                                      _lib.current_stream(self.device))
        _lib.check(rc, "mg_walker_step")
        return self._obs, self._reward, self._done, {"rewards": self._rewards5, "steps": self.steps}

    def set_external_wrench(self, wrench):
        """A push on the base body during the first sub-step of every following launch (mg_walker_params.ext_wrench): float64
        `[6, num_envs]` — force and application point in the base body frame — or None. The tensor is read at launch time, so
        it can be rewritten in place between steps (also inside a captured hipGraph)."""
        if wrench is not None:
            assert wrench.shape == (6, self.num_envs) and wrench.dtype == torch.float64 and wrench.is_contiguous()
        self.ext_wrench = wrench
        self._params_c.ext_wrench = None if wrench is None else wrench.data_ptr()

    def set_terrain(self, boxes):
        """Static boxes on top of the ground plane, shared by every env (mg_walker_params.terrain; wave mapping only) — e.g. the
        list `metagym_amd.quadrupedal.terrain.task_terrain(task)` returns. Each box: (half_extents[3], position[3],
        quaternion (x, y, z, w), friction); the contact's coefficient is friction x the robot's geom friction (Bullet
        multiplies the two bodies' lateral frictions). `None` / [] removes the terrain."""
        self._terrain_spec = list(boxes) if boxes else None
        self._terrain_t = None                     # (a terrain table, if one was installed, goes too)
        if getattr(self, "_robot_set", False):
            self._apply_terrain()

    def set_terrain_table(self, n_courses, max_boxes, terrain_id=None):
        """Per-robot terrains (mg_walker_params.terrain_id): a table of `n_courses` courses of up to `max_boxes` boxes each and the
        course every robot stands on (`terrain_id`: int32 [num_envs] device tensor, default all 0; kept by reference — rewrite it
        in place, e.g. for the robots of a masked reset). Courses start empty; fill them with `write_course`."""
        assert getattr(self, "_robot_set", False), "set_task first"
        N, dev = self.num_envs, self.device
        self._terrain_spec = None
        tab = torch.zeros(int(n_courses), int(max_boxes), _lib.WALKER_BOX_DOUBLES, dtype=torch.float64, device=dev)
        tab[:, :, 0] = 1.0e30                                    # an unused box: zero extent, parked far away (skipped by its x range)
        tab[:, :, 3], tab[:, :, 7], tab[:, :, 11] = 1.0, 1.0, 1.0
        self._terrain_t = tab
        self.terrain_id = torch.zeros(N, dtype=torch.int32, device=dev) if terrain_id is None else terrain_id
        assert self.terrain_id.dtype == torch.int32 and self.terrain_id.shape == (N,) and self.terrain_id.is_contiguous()
        p = self._params_c
        p.n_terrain_boxes, p.terrain, p.terrain_id, p.n_terrain_tables = int(max_boxes), tab.data_ptr(), self.terrain_id.data_ptr(), int(n_courses)

    def write_course(self, index, boxes):
        """Course `index` of the terrain table <- `boxes` ((half_extents, position, quaternion (x, y, z, w), friction) each, like
        set_terrain; fewer than the table's max_boxes: the rest stay unused)."""
        tab = self._terrain_t
        assert tab is not None and tab.dim() == 3 and 0 <= index < tab.shape[0] and len(boxes) <= tab.shape[1], \
            "course %d with %d boxes does not fit the terrain table %s" % (index, len(boxes), None if tab is None else tuple(tab.shape))
        rows = np.zeros((tab.shape[1], _lib.WALKER_BOX_DOUBLES))
        rows[:, 0], rows[:, 3], rows[:, 7], rows[:, 11] = 1.0e30, 1.0, 1.0, 1.0
        if boxes:
            rows[:len(boxes)] = self._box_rows(boxes)
        tab[index].copy_(torch.as_tensor(rows, dtype=torch.float64, device=self.device))

    def _box_rows(self, spec):
        rows = np.zeros((len(spec), _lib.WALKER_BOX_DOUBLES))
        for i, (half, pos, quat, friction) in enumerate(spec):
            x, y, z, w = [float(v) for v in quat]
            nq = np.sqrt(x * x + y * y + z * z + w * w)
            x, y, z, w = x / nq, y / nq, z / nq, w / nq
            R = [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z), 1 - 2 * (x * x + z * z),
                 2 * (y * z - w * x), 2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]
            rows[i, 0:3], rows[i, 3:12], rows[i, 12:15] = pos, R, half
            rows[i, 15] = float(friction) * (1.0 if self.per_proxy_friction else float(self._geom_friction))
        return rows

    def _apply_terrain(self):
        spec, p = getattr(self, "_terrain_spec", None), self._params_c
        tab = getattr(self, "_terrain_t", None)
        if spec is None and tab is not None and tab.dim() == 3:
            # a per-robot terrain table (set_terrain_table / a restored checkpoint) survives set_task(): re-point the new params
            p.n_terrain_boxes, p.terrain = int(tab.shape[1]), tab.data_ptr()
            p.terrain_id, p.n_terrain_tables = self.terrain_id.data_ptr(), int(tab.shape[0])
            return
        p.terrain_id, p.n_terrain_tables = None, 0
        if isinstance(spec, torch.Tensor):          # box rows restored from a checkpoint (load_state_dict): installed as they are
            self._terrain_t = spec.to(device=self.device, dtype=torch.float64).contiguous().clone()
            p.n_terrain_boxes, p.terrain = int(self._terrain_t.shape[0]), self._terrain_t.data_ptr()
            return
        if not spec:
            self._terrain_t, p.n_terrain_boxes, p.terrain = None, 0, None
            return
        rows = np.zeros((len(spec), _lib.WALKER_BOX_DOUBLES))
        for i, (half, pos, quat, friction) in enumerate(spec):
            x, y, z, w = [float(v) for v in quat]
            nq = np.sqrt(x * x + y * y + z * z + w * w)
            x, y, z, w = x / nq, y / nq, z / nq, w / nq
            R = [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z), 1 - 2 * (x * x + z * z),
                 2 * (y * z - w * x), 2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]
            rows[i, 0:3], rows[i, 3:12], rows[i, 12:15] = pos, R, half
            rows[i, 15] = float(friction) * (1.0 if self.per_proxy_friction else float(self._geom_friction))
        self._terrain_t = torch.as_tensor(rows, dtype=torch.float64, device=self.device).contiguous()
        p.n_terrain_boxes, p.terrain = len(spec), self._terrain_t.data_ptr()

    def step_actuated(self, command, kp=None, kd=None, strength=None, limit=None, raw_torque=False, n_substeps=None, log=None,
                      mode="position", kp_env=None, kd_env=None):
        """The engine's in-launch actuators (mg_walker_params.actuation): `n_substeps` physics sub-steps in ONE launch, the
        joint torques re-evaluated before each of them — position control with the reference's PD motor model
        (quadrupedal/robots/laikago_motor.py:136-168) on the current joint state and the desired angles `command`
        (float64 `[n_joints, num_envs]`, SoA), or `raw_torque=True`: `command` are the torques themselves. `mode="hybrid"`
        (laikago_motor.py:143-153; `command` is `[5 n_joints, num_envs]`: desired angle, kp, desired rate, kd, extra torque per
        motor) and `mode="torque"` (:125-128: strength x command, no limit) are the model's other two modes; `kp_env` / `kd_env`
        (`[n_joints, num_envs]`) are per-robot position gains. `log` (float64
        `[n_substeps, 3 n_joints + 7, num_envs]`) receives one true observation per sub-step (joint angles, rates, torques,
        base quaternion x y z w, body-frame angular velocity). The MetaLocomotion outputs (obs / reward / done) of the launch
        are those of the walker rules and are returned for completeness."""
        nj, p = self.n_joints, self._params_c
        cmd = torch.as_tensor(command, dtype=torch.float64, device=self.device)
        assert mode in ("position", "hybrid", "torque")
        rows = 5 * nj if (mode == "hybrid" and not raw_torque) else nj
        assert cmd.shape == (rows, self.num_envs) and cmd.is_contiguous(), "command must be a contiguous [%d, num_envs] float64 tensor" % rows
        k = self.frame_skip if n_substeps is None else int(n_substeps)
        if log is not None:
            assert log.shape == (k, 3 * nj + 7, self.num_envs) and log.is_contiguous() and log.dtype == torch.float64
        saved = (p.frame_skip, p.actuation)
        p.frame_skip, p.pd_command = k, cmd.data_ptr()
        p.actuation = 2 if raw_torque else {"position": 1, "hybrid": 3, "torque": 4}[mode]
        for name, t in (("pd_kp_env", kp_env), ("pd_kd_env", kd_env)):
            if t is not None and not raw_torque:
                assert t.shape == (nj, self.num_envs) and t.is_contiguous() and t.dtype == torch.float64
            setattr(p, name, t.data_ptr() if (t is not None and not raw_torque) else None)
        p.substep_log = None if log is None else log.data_ptr()
        if not raw_torque:
            for name, v, dflt in (("pd_kp", kp, 0.0), ("pd_kd", kd, 0.0), ("pd_strength", strength, 1.0), ("pd_limit", limit, 1e30)):
                getattr(p, name)[:nj] = list(np.broadcast_to(np.asarray(dflt if v is None else v, dtype=np.float64), (nj,)))
        p.step_index = self.global_step
        self.global_step += 1
        try:
            rc = self._lib.mg_walker_step(self._topo, self._models_c, p, self.num_envs, self._state_c, None, _lib.ptr(self._obs),
                                          _lib.ptr(self._reward), _lib.ptr(self._rewards5), _lib.ptr(self._done),
                                          _lib.current_stream(self.device))
        finally:
            p.frame_skip, p.actuation, p.pd_command, p.substep_log = saved[0], saved[1], None, None
            p.pd_kp_env, p.pd_kd_env = None, None
        _lib.check(rc, "mg_walker_step (actuated)")
        return self._obs, self._reward, self._done, {"rewards": self._rewards5, "steps": self.steps}

    def render(self, mode="human", close=False):
        raise NotImplementedError("rendering is out of scope for the batched engine")

    def close(self):
        pass


class MetaHumanoidEnv(WalkerBatchEnv):
    """meta_humanoids/humanoids.py:6-57: 17 actions / 44 observations."""
    robot_dir, variant_prefix = "humanoids", "humanoid"
    foot_list = ("right_foot", "left_foot")
    power = 0.41
    motor_power = [100, 100, 100, 100, 100, 300, 200, 100, 100, 300, 200, 75, 75, 75, 75, 75, 75]   # humanoids.py:19-28
    alive_z, alive_bonus = 0.50, 2.0      # humanoids.py:56
    initial_z = 0.8                       # humanoids.py:48
    torque_f32 = True                     # humanoids.py:50-54: python floats x np.float32 action -> float32 product


class MetaAntEnv(WalkerBatchEnv):
    """meta_ants/ant.py:6-14: 8 actions / 28 observations."""
    robot_dir, variant_prefix = "ants", "ant"
    foot_list = ("front_left_foot", "front_right_foot", "left_back_foot", "right_back_foot")
    power = 2.5
    motor_power = None
    alive_z, alive_bonus = 0.26, 1.0      # ant.py:13
    initial_z = None
