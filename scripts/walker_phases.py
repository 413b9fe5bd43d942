"""Phase profile of the walker wave kernel. Needs a -DMG_WALKER_PROFILE build:
    scripts/build_variant.sh prof WORK -DMG_WALKER_PROFILE
    METAGYM_HIP_LIB=$PWD/metagym_amd/lib/variants/prof.so python scripts/walker_phases.py [humanoid|ant] [n_envs]
Prints the share of the shader-clock cycles each phase of the sub-step takes (summed over all waves)."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from metagym_amd import _lib
from metagym_amd.metalocomotion import MetaAntEnv, MetaHumanoidEnv, variants

NAMES = ["kinematics", "body inertia + subtree sums", "S_d, F_d, h and M entries", "Cholesky", "free motion",
         "detection", "contact rows", "whitening", "PGS", "back-solve + integrate"]
robot = sys.argv[1] if len(sys.argv) > 1 else "humanoid"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
lib = _lib.load()
buf = (ctypes.c_ulonglong * 16)()
substeps = 4
if robot == "a1":       # quadrupedal-v0 from the demo URDF: 13 sub-steps per launch, 23 PGS iterations, optional task=<terrain>
    import numpy as np
    import metagym_amd
    task = sys.argv[3] if len(sys.argv) > 3 else "plane"
    w = np.tile([[0.03], [0.0], [0.02]], (1, 20)) * np.sin(np.linspace(0, 2 * np.pi, 20))
    env = metagym_amd.make("quadrupedal-v0", num_envs=n, urdf="examples/a1_like/a1_like.urdf", device="cuda:0", ETG=1, ETG_w=w,
                           ETG_b=np.zeros(3), auto_reset=True, task=task)
    env.reset()
    a0 = torch.zeros(n, 12, dtype=torch.float64, device="cuda:0")
    acts = [a0] * 8
    substeps = 13
else:
    cls = MetaHumanoidEnv if robot == "humanoid" else MetaAntEnv
    env = cls(num_envs=n, device="cuda:0")
    env.set_task(variants.models(robot, "TRAIN"))
    env.reset(seed=0)
    acts = [torch.rand(n, env.n_joints, device="cuda:0") * 2 - 1 for _ in range(8)]
for i in range(10):
    env.step(acts[i % 8])
torch.cuda.synchronize()
lib.mg_walker_profile_read(buf, 1)
steps = 20
for i in range(steps):
    env.step(acts[i % 8])
torch.cuda.synchronize()
lib.mg_walker_profile_read(buf, 0)
tot = sum(buf[i] for i in range(10))
out = {NAMES[i]: round(100.0 * buf[i] / tot, 1) for i in range(10)}
out["kinematics level loop (of kinematics, incl. the calc_state pass)"] = round(100.0 * buf[11] / tot, 1)
out["cycles_per_substep_per_wave"] = round(tot / (steps * n * substeps))
out["substeps_share_of_kernel_body"] = round(tot / buf[15], 3)
print(json.dumps({"robot": robot, "envs": n, "phase_percent": out}))
