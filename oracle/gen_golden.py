#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the UNMODIFIED reference.

TEST INFRASTRUCTURE (oracle side). Runs only in the build container, where the reference is
mounted at /root/reference; the GPU box never runs this (it has no /root/reference) — it reads the
committed .npz fixtures instead.

    python oracle/gen_golden.py            # regenerate every fixture
    python oracle/gen_golden.py quadrotor  # one family

The reference is imported through the shims in oracle/refstubs (gym / numba / pygame are not
installed here, see oracle/refstubs/README.md) plus two removed-alias shims (np.int, np.product).
numpy.__version__ is recorded in every file: the reference's float choreography depends on NumPy's
promotion rules (NEP 50 under numpy>=2), and this container's numpy is the only runnable definition.

What is recorded
  quadrotor_traj_*.npz   reference `Quadrotor(task='hovering_control')` (quadrotor/env.py:30):
                         reset noise, the f32 action stream, and per env-step the full simulator
                         state (quadrotorsim.py:20-28), obs (env.py:193-209), reward (env.py:211-246),
                         done, power, and whether `_check_failure` (quadrotorsim.py:212-221) raised.
  quadrotor_onestep.npz  256 random (state, action) pairs pushed through ONE `env.step`
                         (10 sub-steps of quadrotorsim.py:122-210) for broad state coverage.
  maze2d_*.npz, maze3d_*.npz: see gen_maze() below.
"""
import os
import sys
import random

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
REF = os.environ.get("METAGYM_REFERENCE", "/root/reference")


def _import_reference():
    if not os.path.isdir(REF):
        raise SystemExit("reference not mounted at %s — golden vectors can only be regenerated "
                         "in the build container" % REF)
    sys.path.insert(0, os.path.join(HERE, "refstubs"))
    sys.path.insert(0, REF)
    np.int = int            # quadrotorsim.py:243,250 uses the removed alias
    np.product = np.prod    # maze_task.py:101 uses the removed alias
    import gym  # noqa: F401  (the stub)
    import metagym.quadrotor  # noqa: F401
    import metagym.metamaze  # noqa: F401
    return gym


# --------------------------------------------------------------------------------------
# Quadrotor
# --------------------------------------------------------------------------------------
def _sim_state(sim):
    return dict(
        pos=np.array(sim.global_position, dtype=np.float32),
        vel=np.array(sim.global_velocity, dtype=np.float64),
        omega=np.array(sim.body_angular_velocity, dtype=np.float64),
        propw=np.array(sim.propeller_angular_velocity, dtype=np.float32),
        R=np.array(sim.rotation_matrix, dtype=np.float32).reshape(9),
    )


def _quadrotor_traj(gym, seed, T, lo, hi, nt=1000, simulator_conf=None):
    env = gym.make("quadrotor-v0", task="hovering_control", nt=nt, simulator_conf=simulator_conf)
    np.random.seed(seed)
    obs0 = env.reset()
    sim = env.simulator
    assert sim.global_velocity.dtype == np.float64 and sim.rotation_matrix.dtype == np.float32
    init = _sim_state(sim)
    actions = np.random.RandomState(seed + 1).uniform(lo, hi, size=(T, 4)).astype(np.float32)
    rec = {k: [] for k in ("pos", "vel", "omega", "propw", "R", "obs", "reward", "done", "power", "ct")}
    failed_at = -1
    for t in range(T):
        try:
            obs, reward, done, info = env.step(actions[t])
        except Exception as e:  # quadrotorsim.py:212-221 raises a bare Exception
            if "quadrotor" not in str(e):
                raise
            failed_at = t
            break
        st = _sim_state(sim)
        for k in ("pos", "vel", "omega", "propw", "R"):
            rec[k].append(st[k])
        rec["obs"].append(np.asarray(obs, dtype=np.float32))
        rec["reward"].append(np.float64(reward))
        rec["done"].append(bool(done))
        rec["power"].append(np.float32(sim.power))
        rec["ct"].append(int(env.ct))
        # NB: like the reference's own tests we keep stepping after `done` without reset();
        # the simulator state simply continues (env.py:144-161 only clears ct).
    out = {"init_" + k: v for k, v in init.items()}
    out["obs0"] = np.asarray(obs0, dtype=np.float32)
    out["actions"] = actions
    for k, v in rec.items():
        out[k] = np.asarray(v)
    out["failed_at"] = np.int64(failed_at)
    out["nt"] = np.int64(nt)
    out["numpy_version"] = np.str_(np.__version__)
    return out


def _quadrotor_onestep(gym, n=256, seed=1234):
    env = gym.make("quadrotor-v0", task="hovering_control")
    env.reset()
    sim = env.simulator
    rs = np.random.RandomState(seed)
    keys = ("pos", "vel", "omega", "propw", "R")
    ins = {k: [] for k in keys}
    outs = {k: [] for k in keys}
    acts, obss, rews, dones, powers = [], [], [], [], []
    for _ in range(n):
        # a random but physically plausible state: small rotation built by the reference's own
        # first-order update (so R is *not* orthonormal, like in a real rollout)
        sim.global_position = (rs.uniform(-20, 20, 3) * [1, 1, 0.2]).astype(np.float32)
        sim.global_velocity = rs.uniform(-5, 5, 3).astype(np.float64)
        sim.body_angular_velocity = rs.uniform(-6, 6, 3).astype(np.float64)
        sim.propeller_angular_velocity = rs.uniform(0, 700, 4).astype(np.float32)
        ang = rs.uniform(-0.6, 0.6, 3)
        cx, sx, cy, sy, cz, sz = np.cos(ang[0]), np.sin(ang[0]), np.cos(ang[1]), np.sin(ang[1]), np.cos(ang[2]), np.sin(ang[2])
        Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
        Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
        Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
        R = (Rz @ Ry @ Rx) * (1.0 + rs.uniform(-2e-3, 2e-3, (3, 3)))
        sim.rotation_matrix = R.astype(np.float32)
        sim._coordination_converter_to_world = sim.rotation_matrix
        sim._coordination_converter_to_body = np.linalg.inv(sim.rotation_matrix)
        env.ct = 0
        st = _sim_state(sim)
        a = rs.uniform(-1.0, 16.0, 4).astype(np.float32)  # includes values outside [0.1, 15]
        obs, reward, done, info = env.step(a)
        so = _sim_state(sim)
        for k in keys:
            ins[k].append(st[k])
            outs[k].append(so[k])
        acts.append(a)
        obss.append(np.asarray(obs, np.float32))
        rews.append(np.float64(reward))
        dones.append(bool(done))
        powers.append(np.float32(sim.power))
    out = {"in_" + k: np.asarray(v) for k, v in ins.items()}
    out.update({"out_" + k: np.asarray(v) for k, v in outs.items()})
    out.update(actions=np.asarray(acts), obs=np.asarray(obss), reward=np.asarray(rews),
               done=np.asarray(dones), power=np.asarray(powers),
               numpy_version=np.str_(np.__version__))
    return out


def _quadrotor_fail(gym):
    """States that trip `_check_failure` (quadrotorsim.py:212-221) inside one env.step: the
    reference raises out of step(); we record the simulator state *at the raise* (the sub-step that
    failed has already been applied, quadrotorsim.py:210)."""
    env = gym.make("quadrotor-v0", task="hovering_control")
    env.reset()
    sim = env.simulator
    cases = [  # pos, vel, omega
        ([0, 0, 999.97], [0, 0, 20.0], [0, 0, 0]),        # leaves the valid zone after a few sub-steps
        ([600, 800, 0.5], [3.0, 4.0, 0], [0.1, 0, 0]),    # |pos| = 1000.0001.. -> first sub-step
        ([0, 0, 0], [150.0, 0, 0], [0, 0, 0]),            # too fast, still >100 after drag
        ([0, 0, 0], [0, 0, 0], [1500.0, 0, 0]),           # angular rate blows through the drag term
        ([0, 0, 0], [1.0, 0, 0], [0, 0, 0.5]),            # healthy control case (must NOT fail)
    ]
    keys = ("pos", "vel", "omega", "propw", "R")
    ins = {k: [] for k in keys}
    outs = {k: [] for k in keys}
    failed, acts = [], []
    for pos, vel, om in cases:
        sim._zero_state()
        sim.global_position = np.array(pos, dtype=np.float32)
        sim.global_velocity = np.array(vel, dtype=np.float64)
        sim.body_angular_velocity = np.array(om, dtype=np.float64)
        sim._coordination_converter_to_world = sim.rotation_matrix
        sim._coordination_converter_to_body = np.linalg.inv(sim.rotation_matrix)
        env.ct = 0
        st = _sim_state(sim)
        a = np.array([3.0, 3.0, 3.0, 3.0], dtype=np.float32)
        try:
            env.step(a)
            failed.append(False)
        except Exception as e:
            if "quadrotor" not in str(e):
                raise
            failed.append(True)
        so = _sim_state(sim)
        for k in keys:
            ins[k].append(st[k])
            outs[k].append(so[k])
        acts.append(a)
    out = {"in_" + k: np.asarray(v) for k, v in ins.items()}
    out.update({"out_" + k: np.asarray(v) for k, v in outs.items()})
    out.update(actions=np.asarray(acts), failed=np.asarray(failed), numpy_version=np.str_(np.__version__))
    return out


def _quadrotor_no_collision(gym):
    """task='no_collision' on the reference's own obstacle map (quadrotor/default_map.txt): pins the
    map path of _check_collision (env.py:248-260: python slicing of the AABB, heights compared with the
    bool np.any) and the no_collision reward branch (env.py:219-221). The map is stored in the file."""
    map_file = os.path.join(REF, "metagym", "quadrotor", "default_map.txt")
    out = {}
    for k, (seed, lo, hi, T) in enumerate([(11, 1.8, 2.6, 250), (12, 0.1, 15.0, 250), (13, 0.1, 1.5, 120)]):
        env = gym.make("quadrotor-v0", task="no_collision", map_file=map_file, nt=200)
        np.random.seed(seed)
        env.reset()
        sim = env.simulator
        if k == 0:
            out["map"] = np.asarray(Quadrotor_map_with_start(env), np.int32)
        init = _sim_state(sim)
        actions = np.random.RandomState(seed + 1).uniform(lo, hi, size=(T, 4)).astype(np.float32)
        rew, done, pos, obs = [], [], [], []
        for t in range(T):
            o, r, d, info = env.step(actions[t])
            rew.append(np.float64(r)); done.append(bool(d)); pos.append(np.array(sim.global_position, np.float32))
            obs.append(np.asarray(o, np.float32))
        out["init_vel_%d" % k], out["init_omega_%d" % k] = init["vel"], init["omega"]
        out["actions_%d" % k], out["reward_%d" % k] = actions, np.asarray(rew)
        out["done_%d" % k], out["pos_%d" % k], out["obs_%d" % k] = np.asarray(done), np.asarray(pos), np.asarray(obs)
        print("no_collision traj", k, "dones", int(np.sum(done)), "first", int(np.argmax(done)) if any(done) else -1)
    out["nt"] = np.int64(200)
    out["numpy_version"] = np.str_(np.__version__)
    return out


def Quadrotor_map_with_start(env):
    """the map as load_map() read it: env.py:109-114 zeroes the start cell after locating it."""
    m = np.array(env.map_matrix)
    m[env.y_offset, env.x_offset] = -1
    return m


def _quadrotor_velocity(gym, seed=7, nt=200, T=260):
    """task='velocity_control': the target trajectory rolled at __init__ in the all-float32 pre-reset
    state (quadrotorsim.py:306-319) and a rollout crossing the ct == nt episode end twice."""
    env = gym.make("quadrotor-v0", task="velocity_control", nt=nt, seed=seed)
    out = {"targets": np.asarray(env.velocity_targets, np.float32), "seed": np.int64(seed), "nt": np.int64(nt)}
    np.random.seed(seed + 100)
    obs0 = env.reset()
    init = _sim_state(env.simulator)
    actions = np.random.RandomState(seed + 1).uniform(1.0, 4.0, size=(T, 4)).astype(np.float32)
    obs, rew, done, ct = [], [], [], []
    for t in range(T):
        o, r, d, info = env.step(actions[t])
        obs.append(np.asarray(o, np.float32)); rew.append(np.float64(r)); done.append(bool(d)); ct.append(int(env.ct))
        assert "next_target_g_v_x" in info
    out.update(obs0=np.asarray(obs0, np.float32), init_vel=init["vel"], init_omega=init["omega"], actions=actions,
               obs=np.asarray(obs), reward=np.asarray(rew), done=np.asarray(done), ct=np.asarray(ct),
               numpy_version=np.str_(np.__version__))
    return out


def gen_quadrotor(gym):
    # (name, seed, T, action range): the SURVEY §8(d) C2 streams — full-range U(0.1,15) and near-hover.
    # seed 3 with nt=50 exercises the `ct == nt` episode end (env.py:159-161).
    specs = [("full_s0", 0, 400, 0.1, 15.0, 1000), ("full_s1", 1, 1000, 0.1, 15.0, 1000),
             ("hover_s2", 2, 400, 2.0, 2.4, 1000), ("short_s3", 3, 120, 1.0, 4.0, 50)]
    for name, seed, T, lo, hi, nt in specs:
        d = _quadrotor_traj(gym, seed, T, lo, hi, nt)
        path = os.path.join(OUT, "quadrotor_traj_%s.npz" % name)
        np.savez_compressed(path, **d)
        print("wrote", path, "steps", len(d["reward"]), "failed_at", int(d["failed_at"]),
              "first done", int(np.argmax(d["done"])) if d["done"].any() else -1)
    # a non-stock simulator config (tests/golden/quadrotor_custom_config.json: off-diagonal inertia,
    # shifted centre of gravity, asymmetric propellers, 3-term thrust polynomial, precision 0.002 ->
    # 5 sub-steps) — pins the general (non-specialised) code path
    d = _quadrotor_traj(gym, 11, 300, 0.2, 14.0, 1000, os.path.join(OUT, "quadrotor_custom_config.json"))
    np.savez_compressed(os.path.join(OUT, "quadrotor_traj_custom_s11.npz"), **d)
    print("wrote quadrotor_traj_custom_s11.npz steps", len(d["reward"]), "failed_at", int(d["failed_at"]),
          "first done", int(np.argmax(d["done"])) if d["done"].any() else -1)
    d = _quadrotor_onestep(gym)
    path = os.path.join(OUT, "quadrotor_onestep.npz")
    np.savez_compressed(path, **d)
    print("wrote", path, "done frac", d["done"].mean())
    d = _quadrotor_velocity(gym)
    np.savez_compressed(os.path.join(OUT, "quadrotor_velocity_control.npz"), **d)
    print("wrote quadrotor_velocity_control.npz dones", int(d["done"].sum()))
    d = _quadrotor_no_collision(gym)
    np.savez_compressed(os.path.join(OUT, "quadrotor_no_collision_map.npz"), **d)
    d = _quadrotor_fail(gym)
    path = os.path.join(OUT, "quadrotor_fail.npz")
    np.savez_compressed(path, **d)
    print("wrote", path, "failed", d["failed"])


def main(argv):
    os.makedirs(OUT, exist_ok=True)
    gym = _import_reference()
    which = set(argv[1:]) or {"quadrotor", "maze"}
    if "quadrotor" in which:
        gen_quadrotor(gym)
    if "maze" in which:
        from gen_golden_maze import gen_maze
        gen_maze(gym)


if __name__ == "__main__":
    main(sys.argv)
