"""quadrupedal-v0 from the demo URDF at 8 192 robots on the reference's terrain tasks (GPU box):  python scripts/quad_a1_terrain_time.py [task ...]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import metagym_amd, bench
from metagym_amd.quadrupedal.terrain import task_terrain
n = 8192
w = np.tile([[0.03], [0.0], [0.02]], (1, 20)) * np.sin(np.linspace(0, 2 * np.pi, 20))
for task in (sys.argv[1:] or ["plane", "stairstair", "slopestair", "balancebeam", "cave"]):
    env = metagym_amd.make("quadrupedal-v0", num_envs=n, urdf="examples/a1_like/a1_like.urdf", device="cuda:0", ETG=1, ETG_w=w, ETG_b=np.zeros(3),
                           auto_reset=True, task=task)
    env.reset()
    a = torch.zeros(n, 12, dtype=torch.float64, device="cuda:0")
    s = bench._time_steps(lambda i: env.step(a), 30, 5)
    print("quadrupedal-v0 8192 robots, task %-12s (%2d terrain boxes): %.3f ms/step eager" % (task, len(task_terrain(task)[2]), s * 1e3), flush=True)
