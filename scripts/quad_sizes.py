"""Quadrotor step time at the headline and north-star batch sizes (GPU box): steady-state batch (bench.QuadrotorShard: staggered
clocks, 1000-step pre-roll, fused auto-reset), HIP events over 200 eager steps.  python scripts/quad_sizes.py [sizes...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

sizes = [int(a) for a in sys.argv[1:]] or [65536, 131072, 1048576]
dev = torch.device("cuda", 0)
out = {"lib": os.environ.get("METAGYM_HIP_LIB", "default")}
for n in sizes:
    q = bench.QuadrotorShard(dev, bench.shard_plan(0, 1, n, "quadrotor"), n)
    s = min(bench._time_steps(q.step, 200, 20) for _ in range(3))
    out[str(n)] = {"us_per_launch": round(s * 1e6, 2), "env_steps_per_s": n / s, "hbm_frac": bench.BYTES_PER_ENV_STEP * n / s / 1e9 / bench.HBM_PEAK_GBS}
    del q
    torch.cuda.empty_cache()
print(json.dumps(out))
