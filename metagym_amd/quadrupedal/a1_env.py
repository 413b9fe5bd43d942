"""`A1GymEnv` — the reference's `quadrupedal-v0` (metagym/quadrupedal/envs/gym_envs/a1_gym_env.py) for N robots, with
everything it computes in Python on the device, composed in the reference's order:

    policy action -> ETGWrapper (+ previous ETG output, MonitorEnv.py:261-273) -> LaikagoPoseOffsetGenerator ->
    [ActionFilter] -> 13 x (ApplyAction -> PHYSICS -> ReceiveObservation) -> info (pose, rot_mat, foot FK, energy) ->
    sensors -> observation[37] -> RewardShaping -> reward, done

The physics is NOT part of this package: `a1/a1.urdf` ships with pybullet_data and the dynamics are PyBullet's, neither is
in the reference tree. `physics` is the caller's batched simulator:

    physics.reset(mask)      -> q[N,12], qd[N,12], base_quat[N,4] (relative to the initial orientation), rpy_rate[N,3] (body
                                frame) after Minitaur.Reset (minitaur.py:398-444)
    physics.substep(torques) -> the same four after one 2 ms pybullet.stepSimulation() with those motor torques
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
from .terrain import task_terrain


class A1GymEnv(object):
    def __init__(self, num_envs, physics, device="cuda:0", ETG=0, ETG_T=0.5, ETG_H=20, ETG_path="", ETG_w=None, ETG_b=None,
                 act_mode="traj", task="plane", normal=0, action_space=0, reward_param=Param_Dict, reward_p=1.0, vel_d=0.6,
                 filter_=0, control_latency=0.002, motor_kp=None, motor_kd=None, env_info=None,
                 motor_control_mode=MotorControlMode.POSITION):
        if physics is None:
            raise _lib.MetaGymHipError(
                "quadrupedal-v0 needs a `physics` object (see metagym_amd/quadrupedal/a1_env.py): the A1 body is not part of this "
                "package — a1/a1.urdf ships with pybullet_data and the dynamics are PyBullet's, neither is in the reference tree. "
                "Everything around the physics (motor model, latency, ETG, sensors, reward) runs on the GPU.")
        self.num_envs, self.device, self.physics = int(num_envs), torch.device(device), physics
        self.task = task
        self.add_height, task_env_info, self.terrain_boxes = task_terrain(task)
        self.env_info = task_env_info if env_info is None else env_info
        self.default_pose = [0.0, 0.0, 0.28 + self.add_height]                      # locomotion_gym_env.py:337
        if hasattr(physics, "set_terrain"):
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
        self.sensors = SensorStack(num_envs, device, normal=normal)
        self.shaping = RewardShaping(num_envs, device, param=reward_param, reward_p=reward_p, vel_d=vel_d, env_info=self.env_info)
        self._lib = _lib.load()
        self.last_torques = None
        self._fusable = motor_control_mode is MotorControlMode.POSITION and motor_kp is None

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

    def _env_step(self, action, reset_mask=None):
        """LocomotionGymEnv.step below the wrappers (locomotion_gym_env.py:461-546)."""
        cmd, etg_obs = self.path.step(action, self.get_time_since_reset())
        if hasattr(self.physics, "fused_step") and self._fusable:      # 13 sub-steps + PD model inside one physics launch
            self.last_torques = self.robot.StepFused(cmd, self.physics.fused_step)
        else:
            self.last_torques = self.robot.Step(cmd, self.physics.substep)
        world, info = self.physics.world(), self._info()
        info.update(base=world["base"], real_contact=world["contact"], bad=world["bad"], real_action=cmd, ETG_obs=etg_obs,
                    ETG_act=self.path.last_ETG_act.t())
        obs = self.sensors.observe(world["base"], info["pose"], info["drpy"], info["joint_angle"], world["contact"], reset_mask)
        return obs, info

    def reset(self):
        """A1GymEnv.reset(): robot.Reset and one observation, sensors reset, ETGWrapper.reset, then the hidden zero-action step
        of RewardShaping.reset (MonitorEnv.py:305-318) whose observation is the one returned."""
        N, d = self.num_envs, self.device
        self.robot.Reset()
        self.robot.ReceiveObservation(*self.physics.reset(None))
        world, info = self.physics.world(), self._info()
        every = torch.ones(N, dtype=torch.bool, device=d)
        self.sensors.observe(world["base"], info["pose"], info["drpy"], info["joint_angle"], world["contact"], every)
        self.path.reset(self.get_time_since_reset())
        obs, _ = self._env_step(torch.zeros(N, 12, dtype=torch.float64, device=d))
        self.shaping.reset(world["base"], info["rot_mat"], info["footposition"])
        info.update(base=world["base"], real_contact=world["contact"])
        return obs, info

    def step(self, action, d_yaw=None):
        obs, info = self._env_step(action)
        reward, done, terms = self.shaping.step(info["base"], info["pose"], info["rot_mat"], info["footposition"],
                                                info["real_contact"], info["energy"], info["bad"], d_yaw)
        info.update(terms)
        return obs, reward, done, info
