"""Pin oracle/maze_sampler.py (MT19937 streams + the Prim digger + numpy's pairwise sum) against tasks
drawn by the unmodified reference sampler: tests/golden/maze_tasks.npz and the task_* fields of the
maze trajectory goldens. CPU-only; bit-exact."""
import glob
import json
import os
import random

import numpy as np
import pytest

from oracle import maze_sampler as ms

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_mt19937_streams_match_python_and_numpy():
    for seed in (0, 1, 7, 12345, 2 ** 31 - 1, 2 ** 40 + 3):
        py = ms.PyRandom(seed)
        random.seed(seed)
        for _ in range(700):     # crosses one state regeneration
            assert py.randbelow(1000) == random.randrange(1000)
        assert py.random() == random.random()
        x, y = list(range(97)), list(range(97))
        py.shuffle(x)
        random.shuffle(y)
        assert x == y
    for seed in (0, 3, 99, 2 ** 32 - 1):
        npr = ms.NpLegacy(seed)
        np.random.seed(seed)
        assert npr.randint(1, 7, 500) == list(np.random.randint(1, 7, size=500))
        assert npr.rand(300) == list(np.random.rand(300))
        assert npr.randint(0, 2, 9) == list(np.random.randint(0, 2, size=9))


def test_pairwise_sum_is_numpys():
    rs = np.random.RandomState(0)
    for _ in range(500):
        n = int(rs.choice([7, 9, 11, 15, 21, 31]))
        a = rs.rand(n, n) * (rs.rand(n, n) < rs.rand()) * float(rs.choice([1.0, 1e-3, 37.0]))
        assert ms.np_sum_f64(a) == float(np.sum(a))


def _same(t, g, k):
    assert tuple(t.start) == tuple(g[k + "start"]) and tuple(t.goal) == tuple(g[k + "goal"])
    assert np.array_equal(t.cell_walls, g[k + "walls"])
    assert np.array_equal(t.cell_texts, g[k + "texts"])
    assert np.array_equal(t.food_rewards, g[k + "food"])          # float64, bit-exact
    assert np.array_equal(t.food_interval, g[k + "interval"])
    assert np.array_equal(np.asarray([t.cell_size, t.wall_height, t.agent_height, t.initial_life, t.max_life,
                                      t.step_reward, t.goal_reward]), g[k + "scalars"])


def test_sampler_matches_reference_tasks():
    g = np.load(os.path.join(GOLDEN, "maze_tasks.npz"))
    cases = json.loads(str(g["cases"]))
    for c, kw in enumerate(cases):
        for seed in g["seeds"]:
            _same(ms.sample_task(int(seed), int(g["n_texts"]), **kw), g, "c%d_s%d_" % (c, seed))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "maze2d_*_s*.npz"))))
def test_sampler_reproduces_trajectory_golden_tasks(path):
    """The maze2d trajectory goldens were recorded on tasks drawn with seed s (file name)."""
    g = np.load(path)
    seed = int(os.path.basename(path)[:-4].rsplit("_s", 1)[1])
    fd = {0: 0.010, 1: 0.05, 2: 0.05}[seed]      # oracle/gen_golden_maze.py
    t = ms.sample_task(seed, 7, n=15, allow_loops=True, crowd_ratio=0.35, step_reward=-0.01, goal_reward=1.0,
                       food_density=fd, food_interval=7)
    assert np.array_equal(t.cell_walls, g["task_cell_walls"])
    assert tuple(t.start) == tuple(int(x) for x in g["task_start"])
    assert tuple(t.goal) == tuple(int(x) for x in g["task_goal"])
    assert np.array_equal(t.food_rewards, g["task_food_rewards"])
    assert np.array_equal(t.food_interval, g["task_food_interval"])
