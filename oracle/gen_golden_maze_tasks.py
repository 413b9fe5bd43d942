#!/usr/bin/env python3
"""Golden maze TASKS drawn by the unmodified reference sampler (maze_task.py:41-190).

TEST INFRASTRUCTURE; runs only in the build container (needs /root/reference). For every
(parameter set, seed) below:   random.seed(seed); numpy.random.seed(seed); MazeTaskSampler(**kw)
and the resulting TaskConfig is stored in tests/golden/maze_tasks.npz. These pin
oracle/maze_sampler.py (CPU) and the device sampler mg_maze_sample_tasks (GPU) bit-exactly.

    python oracle/gen_golden_maze_tasks.py
"""
import json
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden  # noqa: E402  (reference import shims)

CASES = [
    dict(n=15, allow_loops=True, crowd_ratio=0.35, step_reward=-0.01, goal_reward=1.0),          # C1
    dict(n=9, allow_loops=False, step_reward=-0.01, goal_reward=1.0, food_density=0.06, food_interval=20),  # C3
    dict(n=15),                                                                                   # all defaults
    dict(n=7, allow_loops=False),
    dict(n=21, allow_loops=True, crowd_ratio=0.15, food_density=0.03),
    dict(n=11, allow_loops=True, crowd_ratio=0.0),
    dict(n=25, allow_loops=False, food_reward=0.3, food_interval=7),
    dict(n=31, allow_loops=True, crowd_ratio=0.5, step_reward=-0.02),
]
SEEDS = list(range(10)) + [12345, 2 ** 31 - 1]


def main():
    gen_golden._import_reference()
    from metagym.metamaze import MazeTaskSampler
    from metagym.metamaze.envs.maze_task import MAZE_TASK_MANAGER
    out = {"n_texts": np.int64(MAZE_TASK_MANAGER.n_texts), "cases": np.str_(json.dumps(CASES)),
           "seeds": np.asarray(SEEDS, np.int64), "numpy_version": np.str_(np.__version__)}
    for c, kw in enumerate(CASES):
        for seed in SEEDS:
            random.seed(seed)
            np.random.seed(seed)
            t = MazeTaskSampler(**kw)
            k = "c%d_s%d_" % (c, seed)
            out[k + "start"] = np.asarray(t.start, np.int32)
            out[k + "goal"] = np.asarray(t.goal, np.int32)
            out[k + "walls"] = np.asarray(t.cell_walls, np.int8)
            out[k + "texts"] = np.asarray(t.cell_texts, np.uint8)
            out[k + "food"] = np.asarray(t.food_rewards, np.float64)
            out[k + "interval"] = np.asarray(t.food_interval, np.int32)
            out[k + "scalars"] = np.asarray([t.cell_size, t.wall_height, t.agent_height, t.initial_life, t.max_life,
                                             t.step_reward, t.goal_reward], np.float64)
    path = os.path.join(gen_golden.OUT, "maze_tasks.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(CASES) * len(SEEDS), "tasks")


if __name__ == "__main__":
    main()
