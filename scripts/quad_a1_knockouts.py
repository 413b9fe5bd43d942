"""A1 step time against the PGS iteration count (GPU box):  python scripts/quad_a1_knockouts.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import metagym_amd, bench
from metagym_amd.quadrupedal.a1_physics import A1Physics
n = 8192
w = np.tile([[0.03], [0.0], [0.02]], (1, 20)) * np.sin(np.linspace(0, 2 * np.pi, 20))
for it in (23, 5, 1):
    phys = A1Physics(n, urdf="examples/a1_like/a1_like.urdf", device="cuda:0", solver_iterations=it)
    env = metagym_amd.make("quadrupedal-v0", num_envs=n, physics=phys, device="cuda:0", ETG=1, ETG_w=w, ETG_b=np.zeros(3), auto_reset=True)
    env.reset()
    a = torch.zeros(n, 12, dtype=torch.float64, device="cuda:0")
    s = bench._time_steps(lambda i: env.step(a), 20, 5)
    print("solver_iterations %2d: %.3f ms/step" % (it, s * 1e3), flush=True)
