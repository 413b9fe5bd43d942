"""Micro-benchmark: quadrotor step time vs sub-steps per step and vs batch size (GPU box only)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import metagym_amd


def timeit(env, acts, steps=200, warm=20):
    for i in range(warm):
        env.step(acts[i % 8])
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps):
        env.step(acts[i % 8])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3


for n in (65536, 131072, 262144, 1048576):
    acts = torch.rand(8, n, 4, device="cuda") * 14.9 + 0.1
    for dt in (0.001, 0.002, 0.005, 0.01):
        env = metagym_amd.make("quadrotor-v0", num_envs=n, task="hovering_control", dt=dt, auto_reset=True)
        env.reset(seed=0)
        us = timeit(env, acts)
        print("n=%d substeps=%d  %.2f us/launch  %.3g env-steps/s" % (n, round(dt / 0.001), us, n / us * 1e6), flush=True)
    env = metagym_amd.make("quadrotor-v0", num_envs=n, task="hovering_control", auto_reset=True)
    env.reset(seed=0)
    T = 16
    a = torch.rand(T, n, 4, device="cuda") * 14.9 + 0.1
    if n <= 262144:
        for _ in range(3):
            env.rollout(a)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            env.rollout(a)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        print("n=%d rollout T=16: %.2f us/env-step-batch" % (n, (t1 - t0) / 160 * 1e6), flush=True)
