"""Resident envs per CU vs step time (GPU box; needs a -DMG_WALKER_LDS_FLOOR build):
    scripts/build_variant.sh ldsfloor WORK -DMG_WALKER_LDS_FLOOR
    for f in 0 40000 60000; do MG_WALKER_LDS_FLOOR=$f METAGYM_HIP_LIB=metagym_amd/lib/variants/ldsfloor.so python scripts/walker_occupancy_probe.py; done
A dynamic-LDS floor of 40 000 B leaves 4 one-wave workgroups per CU (one wave per SIMD), 60 000 B two."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import metagym_amd, bench
from metagym_amd.metalocomotion import MetaHumanoidEnv, MetaAntEnv, variants
floor = os.environ.get("MG_WALKER_LDS_FLOOR", "0")
n = 8192
w = np.tile([[0.03], [0.0], [0.02]], (1, 20)) * np.sin(np.linspace(0, 2 * np.pi, 20))
env = metagym_amd.make("quadrupedal-v0", num_envs=n, urdf="examples/a1_like/a1_like.urdf", device="cuda:0", ETG=1, ETG_w=w, ETG_b=np.zeros(3), auto_reset=True)
env.reset()
a = torch.zeros(n, 12, dtype=torch.float64, device="cuda:0")
s = bench._time_steps(lambda i: env.step(a), 20, 5)
print("LDS floor %6s B: quadrupedal-v0 8192 robots %.3f ms/step" % (floor, s * 1e3), flush=True)
for cls, robot in ((MetaHumanoidEnv, "humanoid"), (MetaAntEnv, "ant")):
    e = cls(num_envs=n, device="cuda:0", auto_reset=True, max_steps=1000, seed=1)
    e.set_task(variants.models(robot, "TRAIN"))
    e.reset(seed=0)
    acts = [torch.rand(n, e.n_joints, device="cuda:0") * 2 - 1 for _ in range(8)]
    for i in range(60):
        e.step(acts[i % 8])
    s = bench._time_steps(lambda i: e.step(acts[i % 8]), 30, 5)
    print("LDS floor %6s B: %s 8192 envs %.3f ms/step" % (floor, robot, s * 1e3), flush=True)
