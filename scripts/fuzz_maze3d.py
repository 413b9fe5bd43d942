"""Differential fuzzer: maze3d_step_kernel (through the C ABI) against the CPU oracle over random view /
task / frame parameters — frame sizes that are not multiples of the wave or slab size, fields of view,
vision ranges, cell / wall / agent sizes (powers of two and not), dense translucent food. Not part of the
test suite (minutes of oracle time); run on the GPU box:

    python scripts/fuzz_maze3d.py [--configs 120] [--seed 0]

Prints one line per config and a final summary; exits non-zero on any discrete-mode pixel mismatch."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import metagym_amd  # noqa: E402
from metagym_amd.metamaze import MAZE_TASK_MANAGER, MazeTaskSampler  # noqa: E402
from oracle import maze as mo  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", type=int, default=120)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--selftest", action="store_true",
                    help="give the oracle a 1 %% different field of view: the run must then REPORT mismatches")
    args = ap.parse_args()
    tex = np.load(os.path.join(ROOT, "tests", "golden", "maze_textures.npz"))
    MAZE_TASK_MANAGER.set_textures(tex["grounds"], tex["ceil"])
    tex_u8 = MAZE_TASK_MANAGER.grounds.astype(np.uint8)
    rs = np.random.RandomState(args.seed)
    tot_bad = tot_px = fails = 0
    for c in range(args.configs):
        n = int(rs.choice([7, 9, 11, 15, 21]))
        cell = float(rs.choice([0.5, 0.75, 1.0, 1.5, 2.0, 3.0, 4.0]))
        wall_h = cell * float(rs.choice([1.0, 1.6, 2.5]))
        agent_h = wall_h * float(rs.choice([0.3, 0.5, 0.7]))
        H, V = int(rs.randint(8, 220)), int(rs.randint(8, 220))
        fov = float(rs.uniform(0.3, 0.85)) * np.pi
        max_vision = float(rs.choice([3.0, 6.0, 12.0, 25.0]))
        continuous = bool(rs.rand() < 0.35)
        task_type = "SURVIVAL" if rs.rand() < 0.6 else "ESCAPE"
        tt = mo.TASK_TYPES[task_type]
        tasks = [MazeTaskSampler(n=n, allow_loops=bool(rs.rand() < 0.5), crowd_ratio=float(rs.uniform(0.1, 0.5)),
                                 cell_size=cell, wall_height=wall_h, agent_height=agent_h, step_reward=-0.01,
                                 goal_reward=1.0, food_density=float(rs.choice([0.0, 0.05, 0.3])),
                                 food_interval=int(rs.randint(2, 8)), seed=int(rs.randint(1 << 30))) for _ in range(3)]
        n_envs = 9
        name = "meta-maze-continuous-3D-v0" if continuous else "meta-maze-discrete-3D-v0"
        env = metagym_amd.make(name, num_envs=n_envs, device="cuda:0", max_steps=50, resolution=(H, V),
                               task_type=task_type)
        env.max_vision_range, env.fol_angle = max_vision, fov
        env.set_task(tasks)
        ids = env.task_id.cpu().numpy()
        otasks = [mo.Task(**t._asdict()) for t in tasks]
        states = [mo.State(otasks[i]) for i in ids]
        for s, i in zip(states, ids):
            mo.reset(otasks[i], tt, s)
        view = mo.View(tex_u8, MAZE_TASK_MANAGER.ceil, H, V, max_vision=max_vision,
                       fov=fov * (1.01 if args.selftest else 1.0))
        ob = env.reset().cpu().numpy()
        bad = px = 0
        for t in range(7):
            for e in range(n_envs):
                ref = mo.observe_3d(otasks[ids[e]], tt, view, states[e], int(continuous))
                bad += int((ob[e] != ref).sum())
                px += ref.size
            if continuous:
                a = np.stack([rs.uniform(-1.2, 1.2, n_envs), rs.uniform(-0.5, 1.2, n_envs)], 1).astype(np.float32)
            else:
                a = rs.choice(4, size=n_envs, p=[0.25, 0.25, 0.1, 0.4])
            obs, rew, done, _ = env.step(torch.as_tensor(a))
            ob = obs.cpu().numpy()
            r64, d = env.reward64.cpu().numpy(), done.cpu().numpy()
            for e in range(n_envs):
                if continuous:
                    r, dd = mo.step_cont3d(otasks[ids[e]], tt, 50, states[e], a[e][0], a[e][1])
                else:
                    r, dd = mo.step_disc3d(otasks[ids[e]], tt, 50, states[e], a[e])
                if r != r64[e] or dd != d[e]:
                    print("  TRANSITION MISMATCH cfg", c, "t", t, "env", e, r, r64[e], dd, d[e])
                    fails += 1
            if d.any():
                env.reset(mask=done)
                for e in np.nonzero(d)[0]:
                    mo.reset(otasks[ids[e]], tt, states[e])
                ob = env._obs.cpu().numpy() if hasattr(env, "_obs") else ob
        tot_bad += bad
        tot_px += px
        ok = bad == 0 if not continuous else bad <= 1e-3 * px
        fails += 0 if ok else 1
        print("cfg %3d n=%2d cell=%.2f wall=%.2f agent=%.2f %3dx%-3d fov=%.2fpi vision=%4.1f %s %-8s mismatched px %d / %d %s"
              % (c, n, cell, wall_h, agent_h, H, V, fov / np.pi, max_vision, "cont" if continuous else "disc",
                 task_type, bad, px, "" if ok else "<-- FAIL"), flush=True)
        del env
    print("total mismatched pixels %d / %d, failing configs %d" % (tot_bad, tot_px, fails))
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
