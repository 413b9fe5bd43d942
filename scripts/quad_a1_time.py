"""Time quadrupedal-v0 from the demo URDF at 8 192 robots (GPU box):  python scripts/quad_a1_time.py"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
import metagym_amd, bench
n=8192
w = np.tile([[0.03], [0.0], [0.02]], (1, 20)) * np.sin(np.linspace(0, 2 * np.pi, 20))
env = metagym_amd.make("quadrupedal-v0", num_envs=n, urdf="examples/a1_like/a1_like.urdf", device="cuda:0", ETG=1, ETG_w=w, ETG_b=np.zeros(3), auto_reset=True)
env.reset()
a = torch.zeros(n, 12, dtype=torch.float64, device="cuda:0")
s = bench._time_steps(lambda i: env.step(a), 20, 5)
print("quadrupedal 8192: %.3f ms/step eager" % (s*1e3))
