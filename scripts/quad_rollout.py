"""Quadrotor K-steps-per-launch (mg_quadrotor_rollout / step_autoreset with n_steps = K): time per env-step batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import metagym_amd

n = 65536
env = metagym_amd.make("quadrotor-v0", num_envs=n, task="hovering_control", auto_reset=True)
env.reset(seed=0)
for T in (1, 4, 16, 64):
    a = torch.rand(T, n, 4, device="cuda") * 14.9 + 0.1
    for _ in range(3):
        env.rollout(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = max(4, 256 // T)
    e0.record()
    for _ in range(reps):
        env.rollout(a)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * T)
    print("n=%d  K=%3d steps per launch: %.2f us per env-step batch  %.3g env-steps/s" % (n, T, us, n / us * 1e6), flush=True)
