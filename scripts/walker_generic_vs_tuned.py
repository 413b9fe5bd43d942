import sys, os, torch
sys.path.insert(0, os.getcwd())
from metagym_amd.metalocomotion import MetaHumanoidEnv, variants
def run(**kw):
    n = 8192
    env = MetaHumanoidEnv(num_envs=n, device="cuda:0", **kw)
    env.set_task(variants.models("humanoid", "TRAIN")); env.reset(seed=0)
    acts = [torch.rand(n, env.n_joints, device="cuda:0") * 2 - 1 for _ in range(8)]
    for i in range(5): _, _, done, _ = env.step(acts[i % 8])
    env.reset(mask=done); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(40):
        _, _, done, _ = env.step(acts[i % 8])
        if i % 10 == 9: env.reset(mask=done)
    e1.record(); torch.cuda.synchronize()
    print(kw, round(e0.elapsed_time(e1) / 40, 4), "ms", flush=True)
run()
run(per_proxy_friction=True)
