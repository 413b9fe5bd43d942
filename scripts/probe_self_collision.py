import sys, time, torch
sys.path.insert(0, '.')
from metagym_amd.metalocomotion import MetaHumanoidEnv, MetaAntEnv, variants
for cls, robot in ((MetaHumanoidEnv, "humanoid"), (MetaAntEnv, "ant")):
    for sc in (True, False):
        env = cls(num_envs=8192, device="cuda:0", self_collision=sc, auto_reset=True)
        env.set_task(variants.models(robot, "TRAIN"))
        env.reset(seed=0)
        acts = [torch.rand(8192, env.n_joints, device="cuda:0") * 2 - 1 for _ in range(8)]
        for i in range(10): env.step(acts[i % 8])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(40): env.step(acts[i % 8])
        torch.cuda.synchronize()
        print(robot, "self_collision", sc, "%.4f ms" % ((time.perf_counter() - t0) / 40 * 1e3))
