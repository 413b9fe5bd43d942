#!/usr/bin/env python3
"""Golden vectors for the Quadrupedal (A1) ACTUATION path, recorded from the UNMODIFIED reference.

TEST INFRASTRUCTURE; runs only in the build container (needs /root/reference):

    python oracle/gen_golden_a1.py

What runs unmodified: `a1.A1` / `minitaur.Minitaur` (quadrupedal/robots/a1.py, minitaur.py) and
`LaikagoMotorModel` (laikago_motor.py), constructed by their own `__init__`, stepped by their own `Step`:
    Step                      minitaur.py:240-255   action repeat loop
    ProcessAction             minitaur.py:1419-1436 action interpolation
    A1.ApplyAction            a1.py:451-463 (+ _ClipMotorCommands :465-483)
    Minitaur.ApplyAction      minitaur.py:906-955
    _GetPDObservation / _GetDelayedObservation   minitaur.py:1205-1232   latency interpolation over the history deque
    convert_to_torque         laikago_motor.py:92-169   POSITION / HYBRID / TORQUE
    ReceiveObservation        minitaur.py:1184-1203     history push, control observation
    GetMotorAngles / Velocities / Torques / BaseRollPitchYawRate / EnergyConsumptionPerControlStep   minitaur.py:755-885
What does NOT exist here: the A1 body. `a1/a1.urdf` lives in pybullet_data, PyBullet is the physics; neither is in the
reference tree. The robot objects therefore talk to `ScriptedBullet` below: a Bullet-client look-alike whose
"robot" is 12 independent joints (a first-order lag driven by the applied torque) under a base that follows a scripted
attitude — enough to drive every line listed above with moving, closed-loop data, and nothing more. Every value the
scripted world hands to the reference (true joint angles / rates, base orientation, body-frame angular rate) is
recorded per sub-step, so the checkers (oracle/a1.py on the CPU, the HIP kernels on the GPU) are fed the same inputs
and must reproduce the reference's outputs: applied torques, the control observation, the sensor getters, the energy.
"""
import collections
import collections.abc
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("METAGYM_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden", "a1_actuation.npz")


def import_reference():
    if not os.path.isdir(REF):
        raise SystemExit("reference not mounted at %s — run in the build container" % REF)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(HERE, "refstubs"))
    sys.path.insert(0, REF)
    np.int = int
    collections.Sequence = collections.abc.Sequence      # laikago_motor.py:50 (python < 3.10 spelling)
    import metagym.quadrupedal  # noqa: F401
    from metagym.quadrupedal.robots import a1, robot_config
    return a1, robot_config


def quat_mul(a, b):   # (x, y, z, w)
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return (aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
            aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz)


def quat_rotate(q, v):
    x, y, z, w = q
    t = (2 * (y * v[2] - z * v[1]), 2 * (z * v[0] - x * v[2]), 2 * (x * v[1] - y * v[0]))
    return (v[0] + w * t[0] + y * t[2] - z * t[1], v[1] + w * t[1] + z * t[0] - x * t[2],
            v[2] + w * t[2] + x * t[1] - y * t[0])


class ScriptedBullet(object):
    """The calls a1.A1 makes on its `pybullet_client`, answered by a scripted 12-joint world (see the module docstring)."""
    TORQUE_CONTROL, VELOCITY_CONTROL = 2, 0
    URDF_USE_SELF_COLLISION = 8
    JOINTS = ["imu_joint"]
    for leg in ("FR", "FL", "RR", "RL"):
        JOINTS += ["%s_hip_joint" % leg, "%s_upper_joint" % leg, "%s_lower_joint" % leg, "%s_toe_fixed" % leg]

    def __init__(self, seed, dt):
        self.rs = np.random.RandomState(seed)
        self.dt = dt
        self.t = 0.0
        self.q = np.zeros(len(self.JOINTS))
        self.qd = np.zeros(len(self.JOINTS))
        self.tau = np.zeros(len(self.JOINTS))
        self.lag = self.rs.uniform(0.02, 0.06, len(self.JOINTS))       # "inertia" of each scripted joint
        self.visc = self.rs.uniform(0.05, 0.4, len(self.JOINTS))
        self.att = self.rs.uniform(0.05, 0.35, 3), self.rs.uniform(1.0, 4.0, 3), self.rs.uniform(0, 6.28, 3)
        self.base_pos = [0.0, 0.0, 0.28]

    # --- construction-time queries -------------------------------------------------------------------
    def loadURDF(self, *a, **k): return 1
    def getNumJoints(self, body): return len(self.JOINTS)
    def getJointInfo(self, body, i): return (i, self.JOINTS[i].encode("UTF-8"))
    def getDynamicsInfo(self, body, link): return (1.0, 0.5, (0.01, 0.01, 0.01))
    def changeDynamics(self, *a, **k): pass
    def setJointMotorControl2(self, *a, **k): pass
    def resetBasePositionAndOrientation(self, *a, **k): pass
    def resetBaseVelocity(self, *a, **k): pass
    def resetJointState(self, body, joint, angle, targetVelocity=0):
        self.q[joint], self.qd[joint] = angle, targetVelocity

    # --- transforms (only ever applied to quantities that are recorded as INPUTS of the checkers) ------
    def getQuaternionFromEuler(self, e):
        r, p, y = e
        cr, sr, cp, sp, cy, sy = math.cos(r / 2), math.sin(r / 2), math.cos(p / 2), math.sin(p / 2), math.cos(y / 2), math.sin(y / 2)
        return (sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy)
    def getEulerFromQuaternion(self, q):
        x, y, z, w = q
        return (math.atan2(2 * (w * x + y * z), 1 - 2 * (x * x + y * y)), math.asin(max(-1.0, min(1.0, 2 * (w * y - z * x)))),
                math.atan2(2 * (w * z + x * y), 1 - 2 * (y * y + z * z)))
    def invertTransform(self, position, orientation):
        qi = (-orientation[0], -orientation[1], -orientation[2], orientation[3])
        return tuple(-c for c in quat_rotate(qi, position)), qi
    def multiplyTransforms(self, positionA, orientationA, positionB, orientationB):
        return tuple(np.add(positionA, quat_rotate(orientationA, positionB))), quat_mul(orientationA, orientationB)

    # --- the per-sub-step interface ---------------------------------------------------------------------
    def _attitude(self, t):
        amp, om, ph = self.att
        return tuple(amp * np.sin(om * t + ph)), tuple(amp * om * np.cos(om * t + ph))
    def getBasePositionAndOrientation(self, body):
        rpy, _ = self._attitude(self.t)
        return tuple(self.base_pos), self.getQuaternionFromEuler(rpy)
    def getBaseVelocity(self, body):
        _, rate = self._attitude(self.t)
        return (0.1, 0.0, 0.0), rate            # (scripted: euler rates stand in for the world-frame angular velocity)
    def getJointStates(self, body, ids):
        return [(self.q[i], self.qd[i], (0.0,) * 6, self.tau[i]) for i in ids]
    def setJointMotorControlArray(self, bodyIndex, jointIndices, controlMode, forces):
        for i, f in zip(jointIndices, forces):
            self.tau[i] = f
    def stepSimulation(self):
        acc = (self.tau - self.visc * self.qd) / self.lag
        self.qd = self.qd + self.dt * acc
        self.q = self.q + self.dt * self.qd
        self.t += self.dt


def run_case(a1, robot_config, name, seed, mode, n_steps, action_repeat=13, control_latency=0.002, pd_latency=0.0,
             interpolate=False, clip=False, kp=None, kd=None, strength=None, torque_limit=None, big=False):
    dt = 0.002
    world = ScriptedBullet(seed, dt)
    robot = a1.A1(pybullet_client=world, time_step=dt, action_repeat=action_repeat, control_latency=control_latency,
                  enable_action_interpolation=interpolate, enable_clip_motor_commands=clip, motor_control_mode=mode,
                  reset_time=-1)
    robot._pd_latency = pd_latency                        # a constructor argument of Minitaur that A1 does not forward
    if kp is not None:
        robot.SetMotorGains(kp, kd)
    if strength is not None:
        robot.SetMotorStrengthRatios(strength)
    if torque_limit is not None:
        robot._motor_model._torque_limits = np.full(12, torque_limit)
    rs = np.random.RandomState(seed + 1000)
    rec = collections.defaultdict(list)
    # wrap the two halves of _StepInternal to record what goes in and out of each
    apply_action, receive = robot.ApplyAction, robot.ReceiveObservation
    def rec_apply(cmd, m):
        rec["command"].append(np.array(cmd, dtype=np.float64))
        t = apply_action(cmd, m)
        rec["torque"].append(np.array(t, dtype=np.float64))
        rec["observed_torque"].append(np.array(robot._observed_motor_torques, dtype=np.float64))
        return t
    def rec_receive():
        receive()
        true = robot._observation_history[0]
        rec["true_obs"].append(np.array(true, dtype=np.float64))          # q[12] qd[12] torque[12] quat[4] rpy_rate[3]
        rec["control_obs"].append(np.array(robot._control_observation, dtype=np.float64))
    robot.ApplyAction, robot.ReceiveObservation = rec_apply, rec_receive
    rec["first_obs"].append(np.array(robot._observation_history[0], dtype=np.float64))   # pushed by __init__ (:226)
    rec["n_history_at_start"].append(len(robot._observation_history))
    base = np.array([0, 0.9, -1.8] * 4)
    for step in range(n_steps):
        if mode is robot_config.MotorControlMode.TORQUE:
            action = rs.uniform(-20, 20, 12)
        elif mode is robot_config.MotorControlMode.HYBRID:
            action = np.zeros(60)
            action[0::5] = base + rs.uniform(-0.3, 0.3, 12)
            action[1::5] = rs.uniform(40, 400 if big else 120, 12)
            action[2::5] = rs.uniform(-1, 1, 12)
            action[3::5] = rs.uniform(0.5, 4, 12)
            action[4::5] = rs.uniform(-3, 3, 12)
        else:
            action = base + rs.uniform(-0.6 if big else -0.25, 0.6 if big else 0.25, 12)
        rec["action"].append(np.array(action, dtype=np.float64))
        robot.Step(action)
        rec["motor_angles"].append(np.array(robot.GetMotorAngles(), dtype=np.float64))
        rec["motor_velocities"].append(np.array(robot.GetMotorVelocities(), dtype=np.float64))
        rec["motor_torques"].append(np.array(robot.GetMotorTorques(), dtype=np.float64))
        rec["rpy_rate"].append(np.array(robot.GetBaseRollPitchYawRate(), dtype=np.float64))
        rec["energy"].append(float(robot.GetEnergyConsumptionPerControlStep()))
    out = {name + "/" + k: np.array(v) for k, v in rec.items()}
    kps, kds = robot.GetMotorGains()
    out[name + "/config"] = np.array([dt, action_repeat, control_latency, pd_latency, float(interpolate), float(clip),
                                      float(mode.value), n_steps], dtype=np.float64)
    out[name + "/kp"] = np.array(robot._motor_model._kp, dtype=np.float64) * np.ones(12)
    out[name + "/kd"] = np.array(robot._motor_model._kd, dtype=np.float64) * np.ones(12)
    out[name + "/strength"] = np.array(robot._motor_model._strength_ratios, dtype=np.float64) * np.ones(12)
    out[name + "/torque_limit"] = np.array(robot._motor_model._torque_limits, dtype=np.float64)
    return out


def main():
    a1, robot_config = import_reference()
    M = robot_config.MotorControlMode
    cases = [
        dict(name="position_default", seed=1, mode=M.POSITION, n_steps=40),
        dict(name="position_latency_interp", seed=2, mode=M.POSITION, n_steps=30, control_latency=0.0137, pd_latency=0.003,
             interpolate=True),
        dict(name="position_long_latency", seed=3, mode=M.POSITION, n_steps=12, control_latency=0.045, pd_latency=0.0041,
             action_repeat=5),
        dict(name="position_saturating_clip", seed=4, mode=M.POSITION, n_steps=25, big=True, clip=True,
             kp=np.linspace(60, 220, 12), kd=np.linspace(0.5, 3.0, 12), strength=np.linspace(0.6, 1.0, 12), torque_limit=20.0),
        dict(name="hybrid", seed=5, mode=M.HYBRID, n_steps=25, control_latency=0.006),
        dict(name="hybrid_saturating", seed=6, mode=M.HYBRID, n_steps=20, big=True, strength=np.linspace(1.0, 0.5, 12)),
        dict(name="torque", seed=7, mode=M.TORQUE, n_steps=20, strength=np.linspace(0.7, 1.0, 12)),
        dict(name="position_history_wrap", seed=8, mode=M.POSITION, n_steps=12, control_latency=0.031, action_repeat=13),
    ]
    out = {"numpy_version": np.array(np.__version__), "cases": np.array([c["name"] for c in cases])}
    for c in cases:
        out.update(run_case(a1, robot_config, **c))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(cases), "cases")


if __name__ == "__main__":
    main()
