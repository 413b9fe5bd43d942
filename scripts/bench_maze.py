"""Secondary benchmark: MetaMaze throughput on one MI355X (BASELINE configs C1-scaled-out and C3).

    python scripts/bench_maze.py [--envs 16384] [--res 256] [--steps 30]

Prints one JSON line per workload with env-steps/s, the average kernel-launch time (HIP events on
the launching stream) and the achieved fraction of the HBM roofline for the algorithmic bytes of
SURVEY.md §8(d): 12*H*V + 64 B per env-step for the 3-D mazes."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import metagym_amd
from metagym_amd.metamaze import MazeTaskSampler


def timed(env, make_action, steps, warm):
    for _ in range(warm):
        env.step(make_action())
    acts = [make_action() for _ in range(8)]
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps):
        env.step(acts[i % 8])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=16384)
    ap.add_argument("--res", type=int, nargs="*", default=[64, 256])
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--skip2d", action="store_true")
    ap.add_argument("--no-u8", action="store_true", help="skip the uint8 fast-path runs (profiling passes)")
    ap.add_argument("--only", choices=("discrete", "continuous"), default=None,
                    help="run one action space only (separate kernel-trace rows for the two maze3d variants)")
    args = ap.parse_args()
    dev = "cuda:0"
    tasks9 = [MazeTaskSampler(n=9, allow_loops=False, step_reward=-0.01, goal_reward=1.0, food_density=0.06,
                              food_interval=20, seed=s) for s in range(64)]
    n = args.envs
    for res in args.res:
        for name, cont in (("meta-maze-discrete-3D-v0", False), ("meta-maze-continuous-3D-v0", True)):
            if args.only is not None and (args.only == "continuous") != cont:
                continue
            for tt in ("SURVIVAL",):
                if res >= 256 and n * res * res * 12 > 40e9:
                    continue
                env = metagym_amd.make(name, num_envs=n, device=dev, max_steps=200, resolution=(res, res),
                                       task_type=tt, auto_reset=True)
                env.set_task(tasks9)
                env.reset()
                if cont:
                    mk = lambda: torch.rand(n, 2, device=dev) * 2 - 1
                else:
                    mk = lambda: torch.randint(0, 4, (n,), device=dev, dtype=torch.int32)
                s = timed(env, mk, args.steps, args.warmup)
                byt = (12 * res * res + 64) * n
                print(json.dumps({"workload": "%s 9x9 %s %dx%d, %d envs" % (name, tt, res, res, n),
                                  "env_steps_per_s": n / s, "avg_launch_ms": s * 1e3,
                                  "roofline": {"bound": "hbm", "achieved_GBs": byt / s / 1e9, "peak_GBs": 8000.0,
                                               "frac": byt / s / 1e9 / 8000.0, "bytes_per_env_step": 12 * res * res + 64}}),
                      flush=True)
                del env
                torch.cuda.empty_cache()
    for res in args.res:      # the uint8 fast path (non-parity): same renderer, 3 bytes per pixel
        if args.no_u8 or n * res * res * 3 > 40e9:
            continue
        env = metagym_amd.make("meta-maze-discrete-3D-v0", num_envs=n, device=dev, max_steps=200,
                               resolution=(res, res), task_type="SURVIVAL", auto_reset=True, obs_dtype=torch.uint8)
        env.set_task(tasks9)
        env.reset()
        s = timed(env, lambda: torch.randint(0, 4, (n,), device=dev, dtype=torch.int32), args.steps, args.warmup)
        print(json.dumps({"workload": "meta-maze-discrete-3D-v0 9x9 SURVIVAL %dx%d uint8 fast path, %d envs" % (res, res, n),
                          "env_steps_per_s": n / s, "avg_launch_ms": s * 1e3}), flush=True)
        del env
        torch.cuda.empty_cache()
    if not args.skip2d:
        tasks15 = [MazeTaskSampler(n=15, allow_loops=True, crowd_ratio=0.35, step_reward=-0.01, goal_reward=1.0,
                                   food_density=0.02, food_interval=20, seed=s) for s in range(64)]
        for tt in ("ESCAPE", "SURVIVAL"):
            n2 = 1 << 20
            env = metagym_amd.make("meta-maze-2D-v0", num_envs=n2, device=dev, max_steps=200, view_grid=1, task_type=tt,
                                   auto_reset=True)
            env.set_task(tasks15)
            env.reset()
            mk = lambda: torch.randint(0, 4, (n2,), device=dev, dtype=torch.int32)
            s = timed(env, mk, 50, 5)
            print(json.dumps({"workload": "meta-maze-2D-v0 15x15 %s view_grid=1, %d envs" % (tt, n2),
                              "env_steps_per_s": n2 / s, "avg_launch_ms": s * 1e3}), flush=True)
        # on-device task generation (mg_maze_sample_tasks, one wave per task, bit-exact vs the reference sampler)
        from metagym_amd.metamaze import MAZE_TASK_MANAGER
        for nt, kw in ((4096, dict(n=15, allow_loops=True, crowd_ratio=0.35, step_reward=-0.01, goal_reward=1.0)),
                       (4096, dict(n=9, allow_loops=False, step_reward=-0.01, goal_reward=1.0, food_density=0.06,
                                   food_interval=20))):
            MAZE_TASK_MANAGER.sample_tasks_device(nt, device=dev, seed=0, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for r in range(3):
                MAZE_TASK_MANAGER.sample_tasks_device(nt, device=dev, seed=r * nt, **kw)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 3
            print(json.dumps({"workload": "mg_maze_sample_tasks %d tasks n=%d allow_loops=%s" % (nt, kw["n"], kw["allow_loops"]),
                              "ms_per_table": ms, "tasks_per_s": nt / ms * 1e3}), flush=True)


if __name__ == "__main__":
    main()
