"""Differential fuzzer: mg_maze_sample_tasks (device, one wave per task) against oracle/maze_sampler.py under
random sampler parameters and seeds (the oracle itself is fuzzed against the live reference by
oracle/fuzz_vs_reference.py). GPU box only.

    python scripts/fuzz_sampler.py [--configs 150] [--seed 0]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from metagym_amd.metamaze import MAZE_TASK_MANAGER
    from oracle import maze_sampler as ms
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", type=int, default=150)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    tex = np.load(os.path.join(ROOT, "tests", "golden", "maze_textures.npz"))
    MAZE_TASK_MANAGER.set_textures(tex["grounds"], tex["ceil"])
    bad = 0
    for c in range(args.configs):
        rs = np.random.RandomState(args.seed * 100003 + c)
        kw = dict(n=int(rs.choice([7, 9, 11, 13, 15, 17, 21, 25])), allow_loops=bool(rs.rand() < 0.5),
                  crowd_ratio=float(rs.choice([0.0, 0.1, 0.25, 0.35, 0.6])), cell_size=float(rs.choice([1.0, 2.0, 1.5])),
                  step_reward=-float(rs.uniform(0.001, 0.05)), goal_reward=None if rs.rand() < 0.5 else float(rs.uniform(0.5, 3)),
                  food_reward=float(rs.uniform(0.1, 1.0)), food_density=float(rs.choice([0.0, 0.01, 0.05, 0.2])),
                  food_interval=int(rs.randint(1, 200)))
        seeds = [int(x) for x in rs.randint(0, 2 ** 32, size=3, dtype=np.uint64)]
        tasks = MAZE_TASK_MANAGER.sample_tasks_device(len(seeds), device="cuda:0", seeds=seeds, **kw).to_task_configs()
        ok = True
        for t, s in zip(tasks, seeds):
            o = ms.sample_task(s, MAZE_TASK_MANAGER.n_texts, **kw)
            ok = ok and (tuple(t.start) == tuple(o.start) and tuple(t.goal) == tuple(o.goal)
                         and np.array_equal(t.cell_walls, o.cell_walls) and np.array_equal(t.cell_texts, o.cell_texts)
                         and np.array_equal(t.food_rewards, o.food_rewards)
                         and np.array_equal(t.food_interval, o.food_interval) and t.goal_reward == o.goal_reward)
        bad += 0 if ok else 1
        if not ok:
            print("cfg", c, kw, seeds, "DIFFERS", flush=True)
    print("device sampler vs oracle: %d / %d random (parameters, 3 seeds) differ" % (bad, args.configs))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
