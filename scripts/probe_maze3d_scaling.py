"""maze3d_step_kernel time vs frame shape at a fixed batch: fits  t = N (F + H c + H V p)  — per-env fixed cost, per-column
cost (DDA + record broadcast), per-pixel cost.   python scripts/probe_maze3d_scaling.py [n_envs]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import metagym_amd
from metagym_amd.metamaze import MazeTaskSampler

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
task_type = sys.argv[2] if len(sys.argv) > 2 else "SURVIVAL"
dev = "cuda:0"
tasks = [MazeTaskSampler(n=9, allow_loops=False, seed=s) for s in range(64)]
rows = []
for (H, V) in [(64, 64), (128, 64), (64, 128), (128, 128), (256, 64), (64, 256), (256, 128), (128, 256), (256, 256)]:
    env = metagym_amd.make("meta-maze-discrete-3D-v0", num_envs=n, device=dev, max_steps=200, resolution=(H, V), task_type=task_type,
                           auto_reset=True)
    env.set_task(tasks, task_ids=torch.arange(n, device=dev, dtype=torch.int32) % 64)
    env.reset()
    acts = [torch.randint(0, 4, (n,), device=dev, dtype=torch.int32) for _ in range(8)]
    for i in range(5):
        env.step(acts[i % 8])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        env.step(acts[i % 8])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    rows.append((H, V, ms))
    print(json.dumps({"H": H, "V": V, "ms": round(ms, 4), "TB_s": round(n * H * V * 12 / ms / 1e9, 3)}), flush=True)
    del env
    torch.cuda.empty_cache()
A = np.array([[1.0, H, H * V] for H, V, _ in rows]) * n
F, c, p = np.linalg.lstsq(A, np.array([r[2] for r in rows]) * 1e6, rcond=None)[0]      # ns
print(json.dumps({"fit_ns": {"per_env": round(F, 3), "per_column": round(c, 4), "per_pixel": round(p, 6)},
                  "share_at_64x64": {"env": round(F / (F + 64 * c + 4096 * p), 3), "columns": round(64 * c / (F + 64 * c + 4096 * p), 3)},
                  "share_at_256x256": {"env": round(F / (F + 256 * c + 65536 * p), 3), "columns": round(256 * c / (F + 256 * c + 65536 * p), 3)}}))
