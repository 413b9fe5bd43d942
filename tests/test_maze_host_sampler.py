"""The product's host-side `MazeTaskSampler` (metagym_amd/metamaze/maze_task.py) against tasks drawn by the
unmodified reference sampler (tests/golden/maze_tasks.npz, written by oracle/gen_golden_maze_tasks.py with
`random.seed(s); numpy.random.seed(s); MazeTaskSampler(**kw)`): every field identical, float64 food values
included — with the global streams seeded like a reference user would, and with the private `seed=` streams.
CPU-only; nothing here touches the GPU or oracle/."""
import glob
import json
import os
import random

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _same(t, g, k):
    assert tuple(t.start) == tuple(g[k + "start"]) and tuple(t.goal) == tuple(g[k + "goal"])
    assert np.array_equal(t.cell_walls, g[k + "walls"])
    assert np.array_equal(t.cell_texts, g[k + "texts"])
    assert np.array_equal(t.food_rewards, g[k + "food"])          # float64, bit-exact
    assert np.array_equal(t.food_interval, g[k + "interval"])
    assert np.array_equal(np.asarray([t.cell_size, t.wall_height, t.agent_height, t.initial_life, t.max_life,
                                      t.step_reward, t.goal_reward]), g[k + "scalars"])


def test_host_sampler_reproduces_reference_tasks():
    from metagym_amd.metamaze import MAZE_TASK_MANAGER, MazeTaskSampler
    g = np.load(os.path.join(GOLDEN, "maze_tasks.npz"))
    assert MAZE_TASK_MANAGER.n_texts == int(g["n_texts"])
    cases = json.loads(str(g["cases"]))
    n = 0
    for c, kw in enumerate(cases):
        for seed in g["seeds"]:
            seed = int(seed)
            random.seed(seed)                      # the reference user's way: both global streams
            np.random.seed(seed)
            _same(MazeTaskSampler(**kw), g, "c%d_s%d_" % (c, seed))
            random.seed(999)                       # private streams: the globals are neither used nor advanced
            np.random.seed(999)
            before = (random.getstate(), np.random.get_state()[1].copy())
            _same(MazeTaskSampler(seed=seed, **kw), g, "c%d_s%d_" % (c, seed))
            assert random.getstate() == before[0] and np.array_equal(np.random.get_state()[1], before[1])
            n += 1
    assert n == 96


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "maze2d_*_s*.npz"))))
def test_host_sampler_reproduces_the_trajectory_golden_tasks(path):
    """The maze2d trajectory goldens were recorded by the reference on tasks drawn with seed s (file name):
    `metagym_amd.make(...); env.set_task(MazeTaskSampler(seed=s, ...))` therefore plays the reference's maze."""
    from metagym_amd.metamaze import MazeTaskSampler
    g = np.load(path)
    seed = int(os.path.basename(path)[:-4].rsplit("_s", 1)[1])
    fd = {0: 0.010, 1: 0.05, 2: 0.05}[seed]      # oracle/gen_golden_maze.py
    t = MazeTaskSampler(n=15, allow_loops=True, crowd_ratio=0.35, step_reward=-0.01, goal_reward=1.0,
                        food_density=fd, food_interval=7, seed=seed)
    assert np.array_equal(t.cell_walls, g["task_cell_walls"])
    assert tuple(t.start) == tuple(int(x) for x in g["task_start"])
    assert tuple(t.goal) == tuple(int(x) for x in g["task_goal"])
    assert np.array_equal(t.food_rewards, g["task_food_rewards"])
    assert np.array_equal(t.food_interval, g["task_food_interval"])


def test_manager_seed_seeds_both_streams():
    from metagym_amd.metamaze import MAZE_TASK_MANAGER, MazeTaskSampler
    MAZE_TASK_MANAGER.seed(5)
    a = MazeTaskSampler(n=9, allow_loops=False)
    b = MazeTaskSampler(n=9, allow_loops=False, seed=5)
    assert np.array_equal(a.cell_walls, b.cell_walls) and np.array_equal(a.food_rewards, b.food_rewards)
    assert a.start == b.start and a.goal == b.goal
