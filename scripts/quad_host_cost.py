"""Host cost of one Quadrotor.step() (Python + ctypes + hipLaunchKernel) against the kernel it launches.
  * 64 envs: the kernel is a few microseconds, the loop is host-bound -> wall per step = what the host needs per launch;
  * 65 536 envs, K = 20 and K = 200 eager regions (best of 7, synchronised on both sides): what a short timed region pays."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import metagym_amd

dev = "cuda:0"
for n in (64, 65536):
    env = metagym_amd.make("quadrotor-v0", num_envs=n, device=dev, task="hovering_control", auto_reset=True, seed=1)
    env.reset(seed=0)
    acts = [torch.rand(n, 4, device=dev) * 14.9 + 0.1 for _ in range(8)]
    for i in range(200):
        env.step(acts[i % 8])
    torch.cuda.synchronize()
    for K in ((2000,) if n == 64 else (20, 200)):
        best = 1e9
        for rep in range(7):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(K):
                env.step(acts[i % 8])
            t_issue = time.perf_counter() - t0
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / K * 1e6)
        print("n=%d K=%d eager: best wall %.2f us/step (last rep: host issue %.2f us/step)" % (n, K, best, t_issue / K * 1e6), flush=True)
