#!/usr/bin/env python3
"""Golden vectors for the MetaLocomotion PYTHON-SIDE rules, recorded from the UNMODIFIED reference.

TEST INFRASTRUCTURE; runs only in the build container (needs /root/reference):

    python oracle/gen_golden_walker_rules.py

The reference's `MetaHumanoidEnv` / `MetaAntEnv` (metalocomotion/envs/meta_humanoids/meta_humanoids_env.py,
meta_ants/meta_ant_env.py) are imported unchanged through oracle/refstubs (gym, and a PyBullet stand-in whose
dynamics are oracle/abd.py — see oracle/refstubs/pybullet/__init__.py for exactly what that world decides on
Bullet's behalf). Everything the reference computes in Python on top of the physics is recorded per step:
observation (walker_base.py:31-64), the five reward terms, their sum, done, steps (walker_base_env.py:43-82),
the feet-contact flags after the step (:57-63), the reset observation / joint noise / potential
(walker_base.py:13-24, env_bases.py:65-82), together with the physics state after every step, so the checkers
(oracle/abd.py's WalkerEnv on CPU, the HIP kernels on the GPU) can be driven with the same noise and actions.

What this pins: L3 apply_action, L5 calc_state, L6 reward / done, L7 reset — the rules. What it does NOT pin: L4,
the physics itself (the stub's dynamics are this repo's own engine; PyBullet is not in the reference tree).

Cases: 4 humanoid + 2 ant body variants; >= 50 steps each; actions f32 U(-1.2, 1.2) like `Box.sample()` hands them
over (exercising the clip); one episode ends by max_steps, one is followed by a second reset() on the same task
(the floor link is part of `robot.parts` from then on, walker_base_env.py:30-31); a long base-humanoid run falls over.
"""
import itertools
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("METAGYM_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden", "walker_rules.npz")

CASES = [   # (env id, task file, reset seeds (one per episode), steps per episode, max_steps, action seed)
    ("meta-humanoid-v0", "humanoid.xml", [11], [140], 2000, 101),
    ("meta-humanoid-v0", "humanoid_var_tra_000.xml", [12, 13], [50, 25], 2000, 102),
    ("meta-humanoid-v0", "humanoid_var_tra_137.xml", [14], [50], 30, 103),
    ("meta-humanoid-v0", "humanoid_var_ood_003.xml", [15], [50], 2000, 104),
    ("meta-ant-v0", "ant.xml", [16], [60], 2000, 105),
    ("meta-ant-v0", "ant_var_tra_005.xml", [17, 18], [50, 20], 2000, 106),
]


def import_reference():
    if not os.path.isdir(REF):
        raise SystemExit("reference not mounted at %s — run in the build container" % REF)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(HERE, "refstubs"))
    sys.path.insert(0, REF)
    np.int = int
    import gym
    import metagym.metalocomotion  # noqa: F401
    return gym


def robot_state(env):
    rb = env._p._world.bodies[env.robot.objects[0]]
    s = rb.state
    return dict(pos=s.pos.copy(), rot=s.rot.copy(), vel=s.v.copy(), omega=s.w.copy(), q=s.q.copy(), qd=s.qd.copy())


def main():
    gym = import_reference()
    from gym.utils import seeding
    out = {"numpy_version": np.str_(np.__version__), "n_cases": np.int64(len(CASES))}
    for c, (env_id, task, seeds, lengths, max_steps, aseed) in enumerate(CASES):
        env = gym.make(env_id, enable_render=False, max_steps=max_steps)
        env.set_task(task)
        rs = np.random.RandomState(aseed)
        k = "case%d_" % c
        out[k + "env_id"], out[k + "task"] = np.str_(env_id), np.str_(task)
        out[k + "max_steps"], out[k + "episode_lengths"] = np.int64(max_steps), np.asarray(lengths, np.int64)
        rec = {n: [] for n in ("obs", "rewards", "reward", "done", "steps", "feet_contact", "actions", "pos", "rot",
                               "vel", "omega", "q", "qd", "potential")}
        resets = {n: [] for n in ("obs", "joint_noise", "potential", "n_parts")}
        for ep, (seed, T) in enumerate(zip(seeds, lengths)):
            seeding.FORCED_SEEDS = itertools.repeat(seed)      # env_bases.py:57-66 reseeds on every reset
            obs0 = env.reset()
            st = robot_state(env)
            resets["obs"].append(np.asarray(obs0, np.float32))
            resets["joint_noise"].append(st["q"])
            resets["potential"].append(float(env.potential))
            resets["n_parts"].append(len(env.robot.parts))
            for t in range(T):
                a = rs.uniform(-1.2, 1.2, env.action_space.shape).astype(np.float32)
                obs, r, done, info = env.step(a)
                st = robot_state(env)
                rec["actions"].append(a)
                rec["obs"].append(np.asarray(obs, np.float32))
                rec["rewards"].append(np.asarray(info["rewards"], np.float64))
                rec["reward"].append(float(r))
                rec["done"].append(bool(done))
                rec["steps"].append(int(info["steps"]))
                rec["feet_contact"].append(np.asarray(env.robot.feet_contact, np.float32).copy())
                rec["potential"].append(float(env.potential))
                for n in ("pos", "rot", "vel", "omega", "q", "qd"):
                    rec[n].append(st[n])
        seeding.FORCED_SEEDS = None
        for n, v in rec.items():
            out[k + n] = np.asarray(v)
        for n, v in resets.items():
            out[k + "reset_" + n] = np.asarray(v)
        out[k + "joint_names"] = np.asarray([j.joint_name for j in env.robot.ordered_joints])
        out[k + "part_names"] = np.asarray(list(env.robot.parts.keys()))
        out[k + "foot_names"] = np.asarray(list(env.robot.foot_list))
        out[k + "initial_z"] = np.float64(env.robot.initial_z)
        d = np.asarray(rec["done"])
        print(task, "steps", len(d), "done at", np.nonzero(d)[0][:5], "alive<0:", int(np.sum(np.asarray(rec["rewards"])[:, 0] < 0)),
              "feet contact steps:", int(np.asarray(rec["feet_contact"]).any(1).sum()),
              "at-limit max", int(round(-10 * np.asarray(rec["rewards"])[:, 3].min())), "parts", resets["n_parts"])
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
