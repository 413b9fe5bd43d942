"""Random-policy rollouts of every env family, the batched counterpart of the reference's smoke scripts
(metagym/quadrotor/tests/test_env.py, metagym/metamaze/test.py, metagym/metalocomotion/test.py).

    python examples/random_rollouts.py [--envs 4096] [--steps 200]

Needs an AMD GPU (gfx950) with PyTorch-ROCm; there is no CPU path."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metagym_amd  # noqa: E402
from metagym_amd.metamaze import MAZE_TASK_MANAGER  # noqa: E402


def run(name, env, make_action, steps):
    env.reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    total_reward, episodes = 0.0, 0
    for _ in range(steps):
        obs, reward, done, info = env.step(make_action())
        total_reward += float(reward.sum())
        episodes += int(done.sum())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-28s %8d envs  %6.0f steps/s  %.3g env-steps/s  mean reward/step %+.4f  episodes ended %d"
          % (name, env.num_envs, steps / dt, env.num_envs * steps / dt, total_reward / (env.num_envs * steps), episodes))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=200)
    args = ap.parse_args()
    n, dev = args.envs, "cuda"

    quad = metagym_amd.make("quadrotor-v0", num_envs=n, device=dev, task="hovering_control", auto_reset=True)
    run("quadrotor-v0 (hovering)", quad, lambda: torch.rand(n, 4, device=dev) * 14.9 + 0.1, args.steps)

    vel = metagym_amd.make("quadrotor-v0", num_envs=n, device=dev, task="velocity_control", nt=200, seed=1, auto_reset=True)
    run("quadrotor-v0 (velocity)", vel, lambda: torch.rand(n, 4, device=dev) * 14.9 + 0.1, args.steps)

    # tasks are drawn on the GPU, bit-identical to the reference sampler seeded with 0, 1, 2, ...
    tasks15 = MAZE_TASK_MANAGER.sample_tasks_device(64, device=dev, seed=0, n=15, allow_loops=True, crowd_ratio=0.35,
                                                    step_reward=-0.01, goal_reward=1.0)
    maze2d = metagym_amd.make("meta-maze-2D-v0", num_envs=n, device=dev, max_steps=200, view_grid=1, task_type="ESCAPE",
                              auto_reset=True)
    maze2d.set_task(tasks15)
    run("meta-maze-2D-v0", maze2d, lambda: torch.randint(0, 4, (n,), device=dev, dtype=torch.int32), args.steps)

    tasks9 = MAZE_TASK_MANAGER.sample_tasks_device(64, device=dev, seed=0, n=9, allow_loops=False, step_reward=-0.01,
                                                   goal_reward=1.0, food_density=0.06, food_interval=20)
    m3 = min(n, 2048)
    maze3d = metagym_amd.make("meta-maze-discrete-3D-v0", num_envs=m3, device=dev, max_steps=200, resolution=(128, 128),
                              task_type="SURVIVAL", auto_reset=True)
    maze3d.set_task(tasks9)
    run("meta-maze-discrete-3D-v0", maze3d, lambda: torch.randint(0, 4, (m3,), device=dev, dtype=torch.int32), args.steps)

    cont = metagym_amd.make("meta-maze-continuous-3D-v0", num_envs=m3, device=dev, max_steps=200, resolution=(128, 128),
                            task_type="SURVIVAL", auto_reset=True)
    cont.set_task(tasks9)
    run("meta-maze-continuous-3D-v0", cont, lambda: torch.rand(m3, 2, device=dev) * 2 - 1, args.steps)

    # the 384 body variants per robot are regenerated from their generator patterns (no asset directory needed);
    # METAGYM_LOCOMOTION_ASSETS=<reference>/metagym/metalocomotion/envs/assets reads the reference's MJCF files instead
    assets = os.environ.get("METAGYM_LOCOMOTION_ASSETS")
    for ident in ("meta-humanoid-v0", "meta-ant-v0"):
        walker = metagym_amd.make(ident, num_envs=m3, device=dev, assets_dir=assets, auto_reset=True)
        walker.set_task([walker.sample_task("TRAIN") for _ in range(16)])       # 16 body variants, env e runs variant e % 16
        run(ident, walker, lambda: torch.rand(m3, walker.n_joints, device=dev) * 2 - 1, args.steps)

    # quadrupedal-v0 from a URDF on the articulated-body engine. METAGYM_A1_URDF=<pybullet_data>/a1/a1.urdf is the reference's
    # robot (it ships with pybullet_data, not with the reference); the default is this repo's A1-shaped demo robot
    urdf = os.environ.get("METAGYM_A1_URDF", os.path.join(os.path.dirname(os.path.abspath(__file__)), "a1_like", "a1_like.urdf"))
    import numpy as np
    w = np.tile([[0.03], [0.0], [0.02]], (1, 20)) * np.sin(np.linspace(0, 2 * np.pi, 20))      # hand-set ETG weights: an open-loop trot
    a1 = metagym_amd.make("quadrupedal-v0", num_envs=m3, device=dev, urdf=urdf, ETG=1, ETG_w=w, ETG_b=np.zeros(3), task="slopestair",
                          auto_reset=True)
    a1.reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    total, ended = 0.0, 0
    for _ in range(args.steps):
        obs, reward, done, info = a1.step(torch.rand(m3, 12, device=dev, dtype=torch.float64) * 0.2 - 0.1)
        total += float(reward.sum())
        ended += int(done.sum())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-28s %8d envs  %6.0f steps/s  %.3g env-steps/s  mean reward/step %+.4f  episodes ended %d  (13 physics sub-steps per step)"
          % ("quadrupedal-v0 (slopestair)", m3, args.steps / dt, m3 * args.steps / dt, total / (m3 * args.steps), ended))


if __name__ == "__main__":
    main()
