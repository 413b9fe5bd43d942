"""Secondary benchmark: MetaLocomotion humanoid throughput (BASELINE config C4: 8 192 envs on one MI355X)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch

from metagym_amd.metalocomotion import MetaHumanoidEnv, MetaAntEnv
from walker_fixtures import load_models

M = load_models()
for cls, names, n in ((MetaHumanoidEnv, ["humanoid", "humanoid_tra_000", "humanoid_tra_137", "humanoid_ood_003"], 8192),
                      (MetaAntEnv, ["ant", "ant_tra_005"], 8192)):
    env = cls(num_envs=n, device="cuda:0")
    env.set_task([M[k] for k in names])
    env.reset(seed=0)
    acts = [torch.rand(n, env.n_joints, device="cuda:0") * 2 - 1 for _ in range(8)]
    for i in range(5):
        _, _, done, _ = env.step(acts[i % 8])
    env.reset(mask=done)      # the first masked reset pays one-time costs (module load, allocator); keep them out of the timing
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    steps = 40
    e0.record()
    for i in range(steps):
        _, _, done, _ = env.step(acts[i % 8])
        if i % 10 == 9:
            env.reset(mask=done)
    e1.record()
    torch.cuda.synchronize()
    s = e0.elapsed_time(e1) * 1e-3 / steps
    obs_dim = env.obs_dim
    byt = (2 * (3 + 9 + 3 + 3 + 2 * env.n_joints) * 8 + env.n_joints * 4 + obs_dim * 4 + 5) * n
    print(json.dumps({"workload": "%s, %d envs, %d variants" % (cls.__name__, n, len(names)),
                      "env_steps_per_s": n / s, "avg_step_ms": s * 1e3,
                      "algorithmic_GBs": byt / s / 1e9}), flush=True)
