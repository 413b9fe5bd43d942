"""Batched Quadrupedal (Unitree A1) — the ACTUATION path of metagym/quadrupedal (robots/minitaur.py, a1.py,
laikago_motor.py): motor model, observation history with latency, sensor getters — and the control-side wrappers of
A1GymEnv.step (envs/env_wrappers/MonitorEnv.py): ETG action path, reward shaping. The A1 body / physics is not built:
a1.urdf ships with pybullet_data and the dynamics are PyBullet's, neither is in the reference tree."""
from .a1_actuators import A1Actuators, MotorControlMode, SoA, INIT_MOTOR_ANGLES, MOTOR_NAMES
from .a1_env import A1GymEnv
from .a1_physics import A1Physics
from .urdf import load_urdf
from .a1_wrappers import ActionFilter, EtgActionPath, RewardShaping, SensorStack, Param_Dict, FLAT_GROUND

__all__ = ["A1GymEnv", "A1Physics", "load_urdf", "A1Actuators", "MotorControlMode", "SoA", "INIT_MOTOR_ANGLES", "MOTOR_NAMES", "ActionFilter", "EtgActionPath", "RewardShaping", "SensorStack",
           "Param_Dict", "FLAT_GROUND"]
