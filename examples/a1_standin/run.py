"""`quadrupedal-v0` closed-loop on the GPU with the stand-in body of this directory (see physics.py for what that means):

    python examples/a1_standin/run.py [num_envs] [steps] [--unfused] [--graph]

Zero policy actions: the motor model holds the default pose (0, 0.9, -1.8) x 4 through its PD loop, 13 sub-steps per env
step; then the same with the ETG's open-loop trot. Prints base height / reward / done fraction and env-steps/s."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch

import metagym_amd
from physics import StandinPhysics

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if len(args) > 0 else 4096
steps = int(args[1]) if len(args) > 1 else 40
for label, kw in (("hold the default pose", dict(ETG=0)),
                  ("hold the pose on the slopestair start platform", dict(ETG=0, task="slopestair")),
                  ("ETG open-loop gait (hand-set weights)", dict(ETG=1, ETG_w=np.tile([[0.03], [0.0], [0.02]], (1, 20)) * np.sin(np.linspace(0, 2 * np.pi, 20)),
                                                                 ETG_b=np.zeros(3)))):
    phys = StandinPhysics(n, fused="--unfused" not in sys.argv)
    env = metagym_amd.make("quadrupedal-v0", num_envs=n, physics=phys, **kw)
    obs, info = env.reset()
    step = env.step
    if "--graph" in sys.argv:          # the whole env step replayed as one hipGraph
        step = env.capture_step()
        obs, info = env.reset()
    a = torch.zeros(n, 12, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dones = 0.0
    for k in range(steps):
        obs, reward, done, info = step(a)
        dones = max(dones, float(done.double().mean()))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    base = info["base"]
    print("%-40s base z %.3f +- %.3f, x %.3f, reward %.3f, max done fraction %.2f, finite %s, %.2f ms / env step = %.2e env-steps/s"
          % (label, float(base[:, 2].mean()), float(base[:, 2].std()), float(base[:, 0].mean()), float(reward.mean()), dones,
             bool(torch.isfinite(obs).all()), dt * 1e3, n / dt))
