"""Batched Quadrupedal (Unitree A1) — the ACTUATION path of metagym/quadrupedal (robots/minitaur.py, a1.py,
laikago_motor.py): motor model, observation history with latency, sensor getters. The A1 body / physics is not built:
a1.urdf ships with pybullet_data and the dynamics are PyBullet's, neither is in the reference tree."""
from .a1_actuators import A1Actuators, MotorControlMode, INIT_MOTOR_ANGLES, MOTOR_NAMES

__all__ = ["A1Actuators", "MotorControlMode", "INIT_MOTOR_ANGLES", "MOTOR_NAMES"]
