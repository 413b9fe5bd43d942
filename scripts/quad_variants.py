"""Time quadrotor step for experimental library builds (METAGYM_HIP_LIB set by the caller)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import metagym_amd

n = int(os.environ.get("QN", "65536"))
acts = torch.rand(8, n, 4, device="cuda") * 14.9 + 0.1
for dt in ([float(sys.argv[1])] if len(sys.argv) > 1 else (0.01, 0.001)):
    env = metagym_amd.make("quadrotor-v0", num_envs=n, task="hovering_control", dt=dt, auto_reset=True)
    env.reset(seed=0)
    best = []
    for rep in range(4):
        for i in range(50):
            env.step(acts[i % 8])
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(500):
            env.step(acts[i % 8])
        e1.record(); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / 500 * 1e3)
    print(os.environ.get("METAGYM_HIP_LIB", "default"), "substeps=%d" % round(dt / 0.001), " ".join("%.2f" % b for b in best), flush=True)
