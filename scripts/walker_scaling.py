"""Latency vs throughput of the walker step: time per launch for batch sizes from one wave per CU (256 envs) up to
the C4 batch (8 192). Flat time up to 2 048 envs (= 8 resident envs per CU x 256 CUs) means one wave's own latency
sets the launch time; time growing with the batch below that point means the CU is issue-bound."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from metagym_amd.metalocomotion import MetaAntEnv, MetaHumanoidEnv, variants

only = sys.argv[1] if len(sys.argv) > 1 else None
for cls, robot in ((MetaHumanoidEnv, "humanoid"), (MetaAntEnv, "ant")):
    if only is not None and only != robot:
        continue
    models = variants.models(robot, "TRAIN")
    out = {}
    for n in (64, 256, 512, 1024, 2048, 3072, 4096, 8192):
        env = cls(num_envs=n, device="cuda:0")
        env.set_task(models)
        env.reset(seed=0)
        acts = [torch.rand(n, env.n_joints, device="cuda:0") * 2 - 1 for _ in range(8)]
        for i in range(5):
            env.step(acts[i % 8])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        steps = 30
        e0.record()
        for i in range(steps):
            env.step(acts[i % 8])
        e1.record()
        torch.cuda.synchronize()
        out[n] = round(e0.elapsed_time(e1) * 1e3 / steps, 1)
    print(json.dumps({"robot": robot, "us_per_launch_by_envs": out}), flush=True)
