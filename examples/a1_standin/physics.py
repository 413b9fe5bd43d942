"""A batched physics for `metagym_amd.quadrupedal.A1GymEnv` built on this repo's articulated-body engine — an EXAMPLE with a
STAND-IN body (a1_standin.xml: only the leg geometry the reference's Python states; masses / inertias / collision shapes are
placeholders). It shows how the `physics` protocol plugs in and lets `quadrupedal-v0` run closed-loop on the GPU; it says
nothing about the reference's dynamics (PyBullet + pybullet_data/a1/a1.urdf, neither in the reference tree).

Two ways to run an env step: `substep` — one 2 ms `mg_walker_step` launch per sub-step with the torques `A1Actuators` computed
(13 x 2 launches per env step), or `fused_step` — all 13 sub-steps in ONE launch with the reference's PD motor model evaluated
inside the engine before every sub-step (mg_walker_params.actuation) and the 13 observations logged for mg_a1_receive_log.
Both give bit-identical results (tests/test_a1_env_gpu.py)."""
import os

import numpy as np
import torch

from metagym_amd.metalocomotion.mjcf import load_mjcf
from metagym_amd.metalocomotion.walker_env import WalkerBatchEnv
from metagym_amd.quadrupedal import INIT_MOTOR_ANGLES, SoA

XML = os.path.join(os.path.dirname(os.path.abspath(__file__)), "a1_standin.xml")
FEET = ("FR_calf", "FL_calf", "RR_calf", "RL_calf")
TORQUE_LIMIT = 33.5                                  # minitaur.py:88


class _StandinWalker(WalkerBatchEnv):
    variant_prefix = None
    foot_list = FEET
    power, motor_power = TORQUE_LIMIT / 100.0, None   # engine torque = 100 * power * clip(action)
    alive_z, alive_bonus = -1.0, 0.0                  # the walker rules of MetaLocomotion are not used here


class StandinPhysics(object):
    def __init__(self, num_envs, device="cuda:0", solver_iterations=23, fused=True):
        # locomotion_gym_env.py:113-114: 300 / 13 = 23 solver iterations; 2 ms steps (locomotion_gym_config.py:18)
        self.env = _StandinWalker(num_envs=num_envs, device=device, frame_skip=1, time_step=0.002, max_steps=2 ** 30,
                                  solver_iterations=solver_iterations, self_collision=False, preset="mujoco")
        self.env.set_task([load_mjcf(XML, foot_names=FEET, preset="mujoco")])      # the stand-in's MJCF means what MuJoCo says it means
        self.n, self.device = int(num_envs), torch.device(device)
        self._init = torch.as_tensor(np.tile(INIT_MOTOR_ANGLES, (self.n, 1)), dtype=torch.float64, device=self.device)
        if fused:                                  # A1GymEnv takes the one-launch path when the physics offers it
            self.fused_step = self._fused_step

    def _state(self):
        e = self.env
        R = e.rot.t().reshape(self.n, 3, 3)
        w = 0.5 * torch.sqrt(torch.clamp(1.0 + R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2], min=1e-12))
        quat = torch.stack([(R[:, 2, 1] - R[:, 1, 2]) / (4 * w), (R[:, 0, 2] - R[:, 2, 0]) / (4 * w),
                            (R[:, 1, 0] - R[:, 0, 1]) / (4 * w), w], dim=1)
        rate = torch.einsum("nij,ni->nj", R, e.omega.t())                      # body-frame angular velocity R^T omega
        return e.q.t().contiguous(), e.qd.t().contiguous(), quat.contiguous(), rate.contiguous()

    def set_terrain(self, boxes, default_pose):
        """A1GymEnv hands over the task's terrain (metagym_amd.quadrupedal.terrain) and the reset pose [x, y, 0.28 + add_height]
        (locomotion_gym_env.py:337): the boxes go to the engine (mg_walker_params.terrain), the robot is reset that much higher."""
        self.env.set_terrain(boxes)
        self._pose_offset = torch.tensor([default_pose[0], default_pose[1], default_pose[2] - 0.28], dtype=torch.float64,
                                         device=self.device).reshape(3, 1)

    def reset(self, mask):
        self.env.reset(mask=mask, joint_noise=self._init)                      # joints at (0, 0.9, -1.8) x 4 (a1.py:71)
        off = getattr(self, "_pose_offset", None)
        if off is not None:
            m = torch.ones(self.n, dtype=torch.bool, device=self.device) if mask is None else torch.as_tensor(mask, device=self.device).bool()
            self.env.pos += off * m.to(torch.float64)
        return self._state()

    def substep(self, torques):
        """One sub-step with the motor torques A1Actuators computed (raw float64 torques, in-launch actuation mode 2); the
        returned observation is the engine's own sub-step log, so the fused path below sees bit-identical numbers."""
        t = torch.as_tensor(torques, dtype=torch.float64, device=self.device)       # [num_envs, 12] (A1Actuators' torque view)
        assert tuple(t.shape) == (self.n, 12)
        t = t.t().contiguous()                                                      # a no-op copy-free view when it is the kernel's own [12][N] buffer
        if not hasattr(self, "_log1"):
            self._log1 = torch.empty(1, 43, self.n, dtype=torch.float64, device=self.device)
        self.env.step_actuated(t, raw_torque=True, n_substeps=1, log=self._log1)
        g = self._log1[0]
        return SoA(g[0:12]), SoA(g[12:24]), SoA(g[36:40]), SoA(g[40:43])        # [k][N] views, declared as such: taken without a copy

    def _fused_step(self, command, actuators):
        """13 sub-steps in one engine launch, the PD motor model evaluated inside it before every sub-step."""
        kp, kd, strength, limit = actuators.motor_model_parameters()
        if not hasattr(self, "_log"):
            self._log = torch.empty(13, 43, self.n, dtype=torch.float64, device=self.device)
        self.env.step_actuated(command, kp, kd, strength, limit, n_substeps=13, log=self._log)
        return self._log

    def world(self):
        e = self.env
        return dict(base=e.pos.t().contiguous(), contact=e.feet_contact.t().to(torch.float64).contiguous(),
                    bad=torch.zeros(self.n, dtype=torch.int32, device=self.device))
