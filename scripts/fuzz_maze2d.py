"""Differential fuzzer: maze2d_step_kernel (through the C ABI) against the CPU oracle — random maze sizes, food
densities / revival intervals, view_grid 1..3, both task types; explicit masked resets on one batch and the fused
auto-reset on a twin batch (the two must agree). GPU box only.

    python scripts/fuzz_maze2d.py [--configs 60] [--seed 0]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import metagym_amd  # noqa: E402
from metagym_amd.metamaze import MazeTaskSampler  # noqa: E402
from oracle import maze as mo  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", type=int, default=60)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    fails = 0
    for c in range(args.configs):
        rs = np.random.RandomState(args.seed * 100003 + c)
        n = int(rs.choice([7, 9, 15, 21]))
        vg = int(rs.randint(1, 4))
        task_type = "SURVIVAL" if rs.rand() < 0.7 else "ESCAPE"
        tt = mo.TASK_TYPES[task_type]
        max_steps = int(rs.choice([6, 25]))
        tasks = [MazeTaskSampler(n=n, allow_loops=bool(rs.rand() < 0.5), crowd_ratio=float(rs.uniform(0.1, 0.5)),
                                 step_reward=-float(rs.choice([0.01, 0.3])), goal_reward=1.0,
                                 food_density=float(rs.choice([0.02, 0.1, 0.3])), food_interval=int(rs.randint(1, 8)),
                                 initial_life=float(rs.choice([1.0, 0.2])), seed=int(rs.randint(1 << 30))) for _ in range(5)]
        N, S = 1500 + int(rs.randint(0, 200)), 48
        mk = lambda ar: metagym_amd.make("meta-maze-2D-v0", num_envs=N, device="cuda:0", max_steps=max_steps, view_grid=vg,
                                         task_type=task_type, auto_reset=ar)
        env, auto = mk(False), mk(True)
        env.set_task(tasks)
        auto.set_task(tasks)
        ids = env.task_id.cpu().numpy()
        sample = rs.choice(N, S, replace=False)
        otasks = [mo.Task(**t._asdict()) for t in tasks]
        states = {e: mo.State(otasks[ids[e]]) for e in sample}
        for e in sample:
            mo.reset(otasks[ids[e]], tt, states[e])
        o1, o2 = env.reset().clone(), auto.reset().clone()
        ok = torch.equal(o1, o2)
        for t in range(30):
            a = rs.randint(0, 4, N).astype(np.int32)
            at = torch.as_tensor(a).cuda()
            obs, rew, done, _ = env.step(at)
            aobs, arew, adone, _ = auto.step(at)
            ok = ok and torch.equal(rew, arew) and torch.equal(done, adone)
            r64, d, ob = env.reward64.cpu().numpy(), done.cpu().numpy(), obs.cpu().numpy()
            for e in sample:
                r, dd = mo.step_2d(otasks[ids[e]], tt, max_steps, states[e], int(a[e]))
                ok = ok and r == r64[e] and dd == bool(d[e])
                ok = ok and np.array_equal(mo.observe_2d(otasks[ids[e]], tt, states[e], vg), ob[e])
            if d.any():
                obs = env.reset(mask=done)
                for e in sample:
                    if d[e]:
                        mo.reset(otasks[ids[e]], tt, states[e])
            ok = ok and torch.equal(obs, aobs)              # explicit masked reset == fused auto-reset
            sa, sb = env.state_dict(), auto.state_dict()
            ok = ok and all(torch.equal(sa[k], sb[k]) for k in sa)
        fails += 0 if ok else 1
        print("cfg %3d n=%2d view_grid=%d %-8s max_steps=%2d envs=%d %s" % (c, n, vg, task_type, max_steps, N,
                                                                          "ok" if ok else "MISMATCH"), flush=True)
        del env, auto
    print("configs with a mismatch: %d / %d" % (fails, args.configs))
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
