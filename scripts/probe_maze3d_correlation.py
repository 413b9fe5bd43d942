import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, metagym_amd
from metagym_amd.metamaze import MazeTaskSampler
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_maze import timed
dev = "cuda:0"; n = 16384; res = 256
tasks9 = [MazeTaskSampler(n=9, allow_loops=False, step_reward=-0.01, goal_reward=1.0, food_density=0.06, food_interval=20, seed=s) for s in range(64)]
def mkenv(name):
    env = metagym_amd.make(name, num_envs=n, device=dev, max_steps=200, resolution=(res, res), task_type="SURVIVAL", auto_reset=True)
    env.set_task(tasks9); env.reset(); return env
env = mkenv("meta-maze-continuous-3D-v0")
zero = torch.zeros(n, 2, device=dev)
print("cont zero actions (start pose)  %.3f ms" % (timed(env, lambda: zero, 20, 3) * 1e3), flush=True)
turn = torch.zeros(n, 2, device=dev); turn[:, 0] = 0.3
print("cont turn only                  %.3f ms" % (timed(env, lambda: turn, 20, 3) * 1e3), flush=True)
print("cont random                     %.3f ms" % (timed(env, lambda: torch.rand(n, 2, device=dev) * 2 - 1, 30, 5) * 1e3), flush=True)
print("cont random (later)             %.3f ms" % (timed(env, lambda: torch.rand(n, 2, device=dev) * 2 - 1, 30, 60) * 1e3), flush=True)
env = mkenv("meta-maze-discrete-3D-v0")
a0 = torch.zeros(n, dtype=torch.int32, device=dev)
print("disc turn-left only             %.3f ms" % (timed(env, lambda: a0, 20, 3) * 1e3), flush=True)
print("disc random                     %.3f ms" % (timed(env, lambda: torch.randint(0, 4, (n,), device=dev, dtype=torch.int32), 30, 5) * 1e3), flush=True)
print("disc random (later)             %.3f ms" % (timed(env, lambda: torch.randint(0, 4, (n,), device=dev, dtype=torch.int32), 30, 60) * 1e3), flush=True)
