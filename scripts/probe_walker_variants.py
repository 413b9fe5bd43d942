"""How the humanoid step time depends on the number of distinct model variants in the task table
(the table row is a few KB per variant; config C4 deals 256 variants round-robin)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from metagym_amd.metalocomotion import MetaHumanoidEnv
from walker_fixtures import load_models

M = load_models()
base = [M[k] for k in ("humanoid", "humanoid_tra_000", "humanoid_tra_137", "humanoid_ood_003")]
n = 8192
MIX = len(sys.argv) > 1 and sys.argv[1] == "mix"   # rows cycle the 4 parsed variants instead of repeating one
for T in (1, 4, 8, 64, 256):
    env = MetaHumanoidEnv(num_envs=n, device="cuda:0")
    env.set_task([base[(i % 4) if MIX else 0] for i in range(T)])      # T table rows

    env.reset(seed=0)
    acts = [torch.rand(n, env.n_joints, device="cuda:0") * 2 - 1 for _ in range(8)]
    for i in range(5):
        _, _, done, _ = env.step(acts[i % 8])
    env.reset(mask=done)      # first masked reset: one-time costs stay out of the timing
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(30):
        _, _, done, _ = env.step(acts[i % 8])
        if i % 10 == 9:
            env.reset(mask=done)
    e1.record(); torch.cuda.synchronize()
    print(json.dumps({"variants": T, "avg_step_ms": e0.elapsed_time(e1) / 30}), flush=True)
