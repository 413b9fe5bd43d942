import sys, time, json, torch
sys.path.insert(0, '.')
import metagym_amd
from metagym_amd.metamaze import MazeTaskSampler
res = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n, dev = (65536 if res <= 64 else 65536 * 64 * 64 // (res * res)), "cuda:0"
tasks = [MazeTaskSampler(n=9, allow_loops=False, seed=s) for s in range(64)]
for tt in ("SURVIVAL", "ESCAPE"):
    env = metagym_amd.make("meta-maze-discrete-3D-v0", num_envs=n, device=dev, max_steps=200, resolution=(res, res), task_type=tt, auto_reset=True)
    env.set_task(tasks, task_ids=torch.arange(n, device=dev, dtype=torch.int32) % 64)
    env.reset()
    acts = [torch.randint(0, 4, (n,), device=dev, dtype=torch.int32) for _ in range(8)]
    for i in range(5): env.step(acts[i % 8])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20): env.step(acts[i % 8])
    torch.cuda.synchronize()
    print(tt, "%dx%d %d envs: %.4f ms" % (res, res, n, (time.perf_counter() - t0) / 20 * 1e3))
