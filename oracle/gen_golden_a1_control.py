#!/usr/bin/env python3
"""Golden vectors for the Quadrupedal (A1) CONTROL-SIDE wrappers, recorded from the UNMODIFIED reference.

TEST INFRASTRUCTURE; runs only in the build container (needs /root/reference):

    python oracle/gen_golden_a1_control.py

Two pure-Python parts of `A1GymEnv.step` (metagym/quadrupedal/envs/gym_envs/a1_gym_env.py:75) sit between the policy
and `robot.Step`, and between the physics and the returned reward. Both are run here exactly as the reference wires
them (EnvWrapper, envs/env_wrappers/MonitorEnv.py:14-25), on top of a scripted inner env instead of the PyBullet one:

 (B) action path:  ETGWrapper.step / reset              MonitorEnv.py:222-273
                   ETG_layer.update2, ETG_model.forward / act_clip   envs/utilities/ETG_model.py:38-55,98-130
                   A1.ComputeMotorAnglesFromFootLocalPosition        robots/a1.py:493-524 (IK :88-102)
                   TrajectoryGeneratorWrapperEnv.step                envs/env_wrappers/trajectory_generator_wrapper_env.py:61-81
                   LaikagoPoseOffsetGenerator.get_action             envs/env_wrappers/simple_openloop.py:144-165
     recorded: the motor command that reaches `LocomotionGymEnv.step`, info["ETG_obs"], info["ETG_act"].
 (C) reward path:  RewardShaping.step / reset / reward_shaping / terminate and its helpers   MonitorEnv.py:275-519
     recorded: the eight reward terms, the returned reward and done flag, for scripted `info` dictionaries
     (base position, attitude, rotation matrix, foot positions, contacts, energy, bad-contact count).
The scripted inner env decides nothing the checkers are graded on: everything it returns is recorded as an INPUT.
"""
import collections
import collections.abc
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("METAGYM_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden", "a1_control.npz")


def import_reference():
    if not os.path.isdir(REF):
        raise SystemExit("reference not mounted at %s — run in the build container" % REF)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(HERE, "refstubs"))
    sys.path.insert(0, REF)
    np.int = int
    collections.Sequence = collections.abc.Sequence
    import metagym.quadrupedal  # noqa: F401
    from metagym.quadrupedal.envs.env_wrappers import MonitorEnv, simple_openloop, trajectory_generator_wrapper_env
    from metagym.quadrupedal.robots import a1
    return MonitorEnv, simple_openloop, trajectory_generator_wrapper_env, a1


class Box(object):
    def __init__(self, n):
        self.high, self.low = np.ones(n), -np.ones(n)
        self.shape = (n,)


class DebugClient(object):
    """`self.render` of a gym.Wrapper is its bound render METHOD, i.e. truthy: RewardShaping draws its direction line on
    every step (MonitorEnv.py:315-316,368-370). Two no-ops cover it."""
    def addUserDebugLine(self, **kw): return 0
    def removeUserDebugItem(self, i): pass


class InnerEnv(object):
    """What the wrappers see below them: a LocomotionGymEnv-shaped object whose step() is scripted."""
    rendering_enabled = False
    render = False
    env_time_step = 13 * 0.002                       # locomotion_gym_env.py:112
    pybullet_client = DebugClient()

    def __init__(self, robot, infos):
        self.robot, self.infos = robot, infos
        self.observation_space, self.action_space = Box(34), Box(12)
        self.k = 0
        self.commands = []
        self.env_step_counter = 0

    def get_time_since_reset(self):
        return self.k * self.env_time_step           # robot.GetTimeSinceReset(): step counter x time step

    def reset(self, **kwargs):
        self.k = 0
        return (np.zeros(34),), dict(self.infos[0], yaw_init=0.0, latency=0.0, footfriction=1.0, basemass=1.0)

    def step(self, action, **kwargs):
        self.commands.append(np.array(action, dtype=np.float64))
        self.k += 1
        self.env_step_counter += 1
        info = dict(self.infos[min(self.k, len(self.infos) - 1)])
        return (np.zeros(34),), 0.0, False, info


class Robot(object):
    """`env.robot` for act_clip: the reference's own IK method on an object with the two attributes it reads."""
    def __init__(self, a1):
        self._foot_link_ids, self.num_legs, self.num_motors = [0, 1, 2, 3], 4, 12
        self._motor_offset, self._motor_direction = a1.JOINT_OFFSETS, a1.JOINT_DIRECTIONS
        self._a1 = a1
        self.bad = 0

    def ComputeMotorAnglesFromFootLocalPosition(self, leg_id, foot_local_position):
        return self._a1.A1.ComputeMotorAnglesFromFootLocalPosition(self, leg_id, foot_local_position)

    def GetBadFootContacts(self):
        return self.bad

    def GetTimeSinceReset(self):
        return 0.0                                   # (LaikagoPoseOffsetGenerator.get_action deletes it, simple_openloop.py:154)


def scripted_infos(rs, n, env_info, mode):
    """Plausible `info` dictionaries of LocomotionGymEnv.step (locomotion_gym_env.py:534-545)."""
    infos = []
    base = np.array([0.0, 0.0, 0.27])
    yaw = 0.0
    for k in range(n + 2):
        v = rs.uniform([-0.3, -0.2, -0.1], [0.9, 0.2, 0.1])
        if mode == "still" and k > 3:
            v = rs.uniform(-1e-4, 1e-4, 3)
        base = base + 0.026 * v
        yaw += rs.uniform(-0.03, 0.05 if mode != "spin" else 0.12)
        rpy = np.array([rs.uniform(-0.3, 0.3), rs.uniform(-0.3, 0.3), yaw])
        if mode == "tumble" and k > 10:
            rpy[0] = 1.2 + 0.1 * k
        cr, sr, cp, sp, cy, sy = np.cos(rpy[0]), np.sin(rpy[0]), np.cos(rpy[1]), np.sin(rpy[1]), np.cos(rpy[2]), np.sin(rpy[2])
        rot = np.array([cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr, sy * cp, sy * sp * sr + cy * cr,
                        sy * sp * cr - cy * sr, -sp, cp * sr, cp * cr])
        foot = np.array([[0.18, -0.13, -0.25], [0.18, 0.13, -0.25], [-0.18, -0.13, -0.25], [-0.18, 0.13, -0.25]]) + rs.uniform(-0.06, 0.06, (4, 3))
        if mode == "feet_up" and k > 12:
            foot[:, 2] += 0.2
        infos.append(dict(base=tuple(base), pose=rpy.copy(), rot_mat=tuple(rot), footposition=foot,
                          real_contact=[bool(b) for b in rs.rand(4) < 0.7], energy=float(rs.uniform(0, 3)),
                          env_info=env_info))
    return infos


def main():
    MonitorEnv, simple_openloop, tgw, a1 = import_reference()
    out = {"numpy_version": np.array(np.__version__)}

    # ---- (B) action path ---------------------------------------------------------------------------------
    b_cases = [dict(name="etg_traj", act_mode="traj", task_mode="normal", wscale=0.05, seed=1, space=0),
               dict(name="etg_traj_gallop", act_mode="traj", task_mode="gallop", wscale=0.05, seed=2, space=0),
               dict(name="etg_traj_unreachable", act_mode="traj", task_mode="normal", wscale=0.6, seed=3, space=0),
               dict(name="etg_pose", act_mode="pose", task_mode="normal", wscale=0.8, seed=4, space=1),
               dict(name="etg_off", act_mode="traj", task_mode="normal", wscale=0.0, seed=5, space=2, etg=0)]
    for c in b_cases:
        rs = np.random.RandomState(c["seed"])
        H = 20
        w, b = rs.uniform(-1, 1, (3, H)) * c["wscale"], rs.uniform(-1, 1, 3) * c["wscale"] * 0.2
        path = "/tmp/_etg_%s.npz" % c["name"]
        np.savez(path, w=w, b=b)
        robot = Robot(a1)
        inner = InnerEnv(robot, [dict()] * 4)
        tg = tgw.TrajectoryGeneratorWrapperEnv(inner, trajectory_generator=simple_openloop.LaikagoPoseOffsetGenerator(
            action_limit=0.75, action_space=c["space"]))                              # env_builder.py:92-99
        tg.reset = lambda **kw: ((np.zeros(34)), dict())                                # (its reset needs the sensor stack)
        env = MonitorEnv.ETGWrapper(env=tg, ETG=c.get("etg", 1), ETG_T=0.5, ETG_path=path, ETG_T2=0.5, ETG_H=H,
                                    act_mode=c["act_mode"], task_mode=c["task_mode"], step_y=0.05)
        obs, info = env.reset()
        rec = collections.defaultdict(list)
        if c.get("etg", 1):
            rec["reset_etg_act"].append(np.array(info["ETG_act"]))
            rec["reset_etg_obs"].append(np.array(info["ETG_obs"]))
        for k in range(40):
            action = rs.uniform(-0.5, 0.5, 12)
            rec["action"].append(action)
            rec["t"].append(inner.get_time_since_reset())
            _, _, _, info = env.step(action)
            rec["command"].append(inner.commands[-1])
            if c.get("etg", 1):
                rec["etg_act"].append(np.array(info["ETG_act"]))
                rec["etg_obs"].append(np.array(info["ETG_obs"]))
        for k2, v in rec.items():
            out[c["name"] + "/" + k2] = np.array(v)
        out[c["name"] + "/w"], out[c["name"] + "/b"] = w, b
        out[c["name"] + "/config"] = np.array([c.get("etg", 1), 0.5, 0.5, H, 0.04, 0.2, c["act_mode"] == "pose",
                                               c["task_mode"] == "gallop", c["space"], inner.env_time_step], dtype=np.float64)
    out["b_cases"] = np.array([c["name"] for c in b_cases])

    # ---- (C) reward path ---------------------------------------------------------------------------------
    flat = [[-100, 100, np.array([1, 0, 0, 0, 0, 0, 0])]]                               # locomotion_gym_env.py:76
    slopes = [[-100, 0.3, np.array([0, 0, 0, 0, 0, 0, 0])], [0.3, 0.8, np.array([1, 0, 0, 0, 0.25, 0, 0])],
              [0.8, 100, np.array([0, 1, 0, 0, -0.2, 0, 0])]]
    c_cases = [dict(name="reward_walk", seed=11, mode="walk", env_info=flat, d_yaw=0.0, n=60),
               dict(name="reward_yaw_target", seed=12, mode="spin", env_info=flat, d_yaw=0.4, n=40),
               dict(name="reward_slopes", seed=13, mode="walk", env_info=slopes, d_yaw=0.0, n=70),
               dict(name="reward_still", seed=14, mode="still", env_info=flat, d_yaw=0.0, n=30),
               dict(name="reward_tumble", seed=15, mode="tumble", env_info=flat, d_yaw=0.0, n=25),
               dict(name="reward_feet_up", seed=16, mode="feet_up", env_info=flat, d_yaw=0.0, n=25,
                    param={'torso': 0.7, 'up': 0.5, 'feet': 0.3, 'tau': 0.02, 'done': 1, 'velx': 0, 'badfoot': 0.2, 'footcontact': 0.15},
                    reward_p=5.0, vel_d=0.4),
               # vel_mode "equal" (MonitorEnv.py:515-518): exp(-5 |v - vel_d|) instead of min(vel_d, v)
               dict(name="reward_walk_equal", seed=17, mode="walk", env_info=flat, d_yaw=0.0, n=50, vel_mode="equal"),
               dict(name="reward_slopes_equal", seed=18, mode="walk", env_info=slopes, d_yaw=0.3, n=60, vel_mode="equal", vel_d=0.45)]
    for c in c_cases:
        rs = np.random.RandomState(c["seed"])
        infos = scripted_infos(rs, c["n"], c["env_info"], c["mode"])
        robot = Robot(a1)
        inner = InnerEnv(robot, infos)
        env = MonitorEnv.RewardShaping(env=inner, param=c.get("param", MonitorEnv.Param_Dict), reward_p=c.get("reward_p", 1.0),
                                       vel_d=c.get("vel_d", 0.6), vel_mode=c.get("vel_mode", "max"))
        kw = dict(d_yaw=c["d_yaw"]) if c["d_yaw"] else {}
        env.reset(**kw)
        rec = collections.defaultdict(list)
        for k in range(c["n"]):
            robot.bad = int(rs.randint(0, 3))
            info_in = infos[min(inner.k + 1, len(infos) - 1)]
            _, reward, done, info = env.step(np.zeros(12), **kw)
            rec["base"].append(np.array(info_in["base"]))
            rec["pose"].append(np.array(info_in["pose"]))
            rec["rot_mat"].append(np.array(info_in["rot_mat"]))
            rec["footposition"].append(np.array(info_in["footposition"]))
            rec["real_contact"].append(np.array(info_in["real_contact"], dtype=np.float64))
            rec["energy"].append(info_in["energy"])
            rec["bad"].append(robot.bad)
            rec["terms"].append(np.array([info[t] for t in ("torso", "up", "feet", "tau", "badfoot", "footcontact")]))
            rec["reward"].append(reward)
            rec["done"].append(bool(done))
            rec["vel"].append(np.array(info["vel"]))
            rec["foot_world"].append(np.array(info["foot_position_world"]))
        for k2, v in rec.items():
            out[c["name"] + "/" + k2] = np.array(v)
        first = infos[0]
        out[c["name"] + "/reset_base"] = np.array(first["base"])
        out[c["name"] + "/reset_foot_world"] = np.array(env.get_foot_world(first))
        p = c.get("param", MonitorEnv.Param_Dict)
        out[c["name"] + "/param"] = np.array([p[t] for t in ("torso", "up", "feet", "tau", "badfoot", "footcontact")], dtype=np.float64)
        out[c["name"] + "/config"] = np.array([c.get("reward_p", 1.0), c.get("vel_d", 0.6), c["d_yaw"]], dtype=np.float64)
        out[c["name"] + "/vel_mode"] = np.array(c.get("vel_mode", "max"))
        out[c["name"] + "/segments"] = np.array([[s[0], s[1], s[2][0], s[2][1], s[2][4]] for s in c["env_info"]], dtype=np.float64)
    out["c_cases"] = np.array([c["name"] for c in c_cases])
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
    for c in c_cases:
        print(c["name"], "done steps:", int(np.sum(out[c["name"] + "/done"])), "reward range", out[c["name"] + "/reward"].min(), out[c["name"] + "/reward"].max())
    for c in b_cases:
        if c["name"] + "/etg_act" in out:
            print(c["name"], "|etg_act| max", np.abs(out[c["name"] + "/etg_act"]).max())


if __name__ == "__main__":
    main()
