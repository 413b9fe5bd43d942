"""Worker functions of bench.py's reference CPU legs, in a module light enough to be imported by a SPAWNED process (numpy only:
bench.py itself imports torch and has a live HIP runtime by the time it times the CPU — forking that process works until the day a
runtime thread holds a lock at the moment of the fork). TEST / MEASUREMENT INFRASTRUCTURE, like everything under oracle/.

Each worker imports the UNMODIFIED reference from `ref` (oracle/_ref, built by oracle/make_ref.py, or a reference tree) through
oracle/refstubs (gym / pygame / numba are not installed) and steps ONE environment object — the reference has no batched mode
("reference env.step in a multiprocessing.Pool(P) of independent single-env workers")."""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))


def _paths(ref):
    for p in (os.path.join(HERE, "refstubs"), ref):
        if p in sys.path:
            sys.path.remove(p)
    sys.path.insert(0, os.path.join(HERE, "refstubs"))
    sys.path.insert(0, ref)


def quadrotor_worker(args):
    """metagym/quadrotor/env.py:127 `Quadrotor.step`, hovering_control, dt = 0.01, nt = 1000, U(0.1, 15) actions, finished episodes
    reset: 20 warm-up steps, then steps until the deadline. -> (steps, seconds)"""
    idx, ref, seconds = args
    import numpy as np
    _paths(ref)
    np.int = int                      # quadrotorsim.py:243,250 use the removed alias
    import gym  # noqa: F401  (the stub)
    from metagym.quadrotor.env import Quadrotor
    np.random.seed(1000 + idx)
    env = Quadrotor(task="hovering_control", nt=1000)
    env.reset()
    rs = np.random.RandomState(2000 + idx)
    acts = rs.uniform(0.1, 15.0, (256, 4)).astype(np.float32)
    n = 0

    def one(k):
        try:
            _, _, done, _ = env.step(acts[k % 256])
        except Exception:             # _check_failure raises out of step() (quadrotorsim.py:212-221)
            done = True
        if done:
            env.reset()
    for k in range(20):
        one(k)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        one(n)
        n += 1
    return n, time.perf_counter() - t0


def maze3d_worker(args):
    """One single-env worker of the unmodified MetaMazeDiscrete3D (maze_discrete_3d.py:44-126; numba is not installed, so this is
    the un-jitted Python the stub runs — "for the record"). -> (steps, seconds)"""
    idx, ref, seconds, res = args
    import random
    import numpy as np
    _paths(ref)
    np.int = int
    np.product = np.prod                # maze_task.py:101 uses the removed alias
    import gym
    import metagym.metamaze  # noqa: F401
    from metagym.metamaze import MazeTaskSampler
    random.seed(idx)
    np.random.seed(idx)
    env = gym.make("meta-maze-discrete-3D-v0", enable_render=False, task_type="SURVIVAL", max_steps=200,
                   resolution=(res, res))
    env.set_task(MazeTaskSampler(n=9, allow_loops=False, step_reward=-0.01, goal_reward=1.0, food_density=0.06,
                                 food_interval=20))
    env.reset()
    rs = np.random.RandomState(idx)
    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        if env.step(int(rs.randint(4)))[2]:
            env.reset()
        n += 1
    return n, time.perf_counter() - t0


def maze2d_worker(args):
    """BASELINE.json configs[0]: ONE unmodified MetaMaze2D (maze_env.py:147-204, `step` at :189-204 ->
    maze_2d.py do_action / get_observation), 15x15, crowd_ratio 0.35, view_grid 1, `task_type` ESCAPE or SURVIVAL, uniform random
    actions, a finished episode is reset (the reset is inside the clock, as it is for the other legs). -> (steps, seconds)"""
    idx, ref, seconds, task_type = args
    import random
    import numpy as np
    _paths(ref)
    np.int = int
    np.product = np.prod                # maze_task.py:101 uses the removed alias
    import gym
    import metagym.metamaze  # noqa: F401
    from metagym.metamaze import MazeTaskSampler
    random.seed(idx)
    np.random.seed(idx)
    env = gym.make("meta-maze-2D-v0", enable_render=False, task_type=task_type, max_steps=200, view_grid=1)
    env.set_task(MazeTaskSampler(n=15, allow_loops=True, crowd_ratio=0.35, step_reward=-0.01, goal_reward=1.0,
                                 food_density=0.06, food_interval=20))
    env.reset()
    rs = np.random.RandomState(idx)
    acts = rs.randint(0, 4, 4096)
    for k in range(200):                # warm-up (imports, first-call paths)
        if env.step(int(acts[k]))[2]:
            env.reset()
    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        if env.step(int(acts[n & 4095]))[2]:
            env.reset()
        n += 1
    return n, time.perf_counter() - t0


if __name__ == "__main__":      # python oracle/ref_workers.py quadrotor|maze3d|maze2d <idx> <ref> <seconds> [<res> | <task_type>]  ->  "<steps> <seconds>"
    kind, idx, ref, seconds = sys.argv[1], int(sys.argv[2]), sys.argv[3], float(sys.argv[4])
    if kind == "quadrotor":
        out = quadrotor_worker((idx, ref, seconds))
    elif kind == "maze2d":
        out = maze2d_worker((idx, ref, seconds, sys.argv[5]))
    else:
        out = maze3d_worker((idx, ref, seconds, int(sys.argv[5])))
    print("%d %.6f" % out, flush=True)
