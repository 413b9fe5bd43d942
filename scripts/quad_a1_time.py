"""Time quadrupedal-v0 from the demo URDF at several batch sizes, eager and as a captured hipGraph (GPU box):
    python scripts/quad_a1_time.py [n ...]           (default 64 512 8192)
Under `rocprofv3 --kernel-trace --stats` with MG_A1_TRACE_STEPS=k it runs k eager env steps at the first size only, so that the
trace's call counts divide by k into launches per env step."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
import metagym_amd, bench
sizes = [int(x) for x in sys.argv[1:]] or [64, 512, 8192]
trace_steps = int(os.environ.get("MG_A1_TRACE_STEPS", "0"))
w = np.tile([[0.03], [0.0], [0.02]], (1, 20)) * np.sin(np.linspace(0, 2 * np.pi, 20))
for n in sizes:
    env = metagym_amd.make("quadrupedal-v0", num_envs=n, urdf="examples/a1_like/a1_like.urdf", device="cuda:0", ETG=1, ETG_w=w, ETG_b=np.zeros(3), auto_reset=True)
    env.reset()
    a = torch.zeros(n, 12, dtype=torch.float64, device="cuda:0")
    if trace_steps:
        for i in range(5):
            env.step(a)
        torch.cuda.synchronize()
        print("TRACE_BEGIN", flush=True)
        for i in range(trace_steps):
            env.step(a)
        torch.cuda.synchronize()
        break
    s = bench._time_steps(lambda i: env.step(a), 40, 5)
    line = "quadrupedal-v0 %5d robots: %.3f ms per env step eager" % (n, s * 1e3)
    try:
        replay = env.capture_step()
        sg = bench._time_steps(lambda i: replay(a), 40, 5)
        line += ", %.3f ms as a captured hipGraph (eager / graph = %.2f)" % (sg * 1e3, s / sg)
    except Exception as e:
        line += " (capture_step failed: %r)" % (e,)
    print(line, flush=True)
    del env
    torch.cuda.empty_cache()
