"""An env built on cuda:1 while cuda:0 is the thread's current device must launch on cuda:1: the C ABI looks up
the device that owns the state memory and selects it for the duration of the call (torch's default
stream handle is 0 on every device, so without the guard the kernels would run on cuda:0 against cuda:1
pointers). Needs two GPUs; skipped on the single-GPU test box."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_env_on_second_device_while_first_is_current():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import metagym_amd
    torch.cuda.set_device(0)
    envs = [metagym_amd.make("quadrotor-v0", num_envs=257, device="cuda:%d" % d, task="hovering_control", nt=6,
                             auto_reset=True, seed=3) for d in (0, 1)]
    for e in envs:
        e.reset(seed=1)
    acts = np.random.RandomState(0).uniform(0.1, 15, (14, 257, 4)).astype(np.float32)
    for t in range(14):
        outs = [e.step(torch.as_tensor(acts[t]).to(e.device)) for e in envs]
        assert torch.cuda.current_device() == 0
        assert torch.equal(outs[0][0].cpu(), outs[1][0].cpu()) and torch.equal(outs[0][2].cpu(), outs[1][2].cpu())
    s0, s1 = envs[0].state_dict(), envs[1].state_dict()
    for k in ("pos", "vel", "omega", "propw", "rot", "ct", "episode"):
        assert torch.equal(s0[k].cpu(), s1[k].cpu()), k
    assert int(s1["episode"].min()) == 2


def _mixed_shard_runs(devices, n=512):
    """bench.MixedStep (the per-GPU step of `bench.py --workload mixed`) for one and the same shard plan on each of `devices`,
    cuda:0 staying the thread's current device: final quadrotor state, last observations, maze frames / rewards / step counters."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    torch.cuda.set_device(0)
    plan = bench.shard_plan(0, 1, n, "mixed")
    runs = []
    for d in devices:
        dev = torch.device("cuda", d)
        quad, maze = bench.QuadrotorShard(dev, plan, n, preroll=8, nt=6), bench.MazeShard(dev, plan, n, res=32, max_steps=12)
        step = bench.MixedStep(dev, quad, maze)
        for i in range(15):
            step(i)
            assert torch.cuda.current_device() == 0
        torch.cuda.synchronize(dev)
        runs.append((quad.env.state_dict(), quad.env._obs.cpu(), maze.env._obs.cpu(), maze.env.reward64.cpu(), maze.env.steps.cpu()))
    return runs


def _assert_same_runs(a, b):
    for k in ("pos", "vel", "omega", "propw", "rot", "ct", "episode"):
        assert torch.equal(a[0][k].cpu(), b[0][k].cpu()), k
    for x, y in zip(a[1:], b[1:]):
        assert torch.equal(x, y)
    assert int(b[0]["episode"].min()) >= 2


def test_mixed_c5_shard_is_reproducible_on_one_device():
    """The harness of the two-device test below, on the one device every box has: the same shard plan built twice gives the same
    trajectories (so a difference between cuda:0 and cuda:1 below can only come from the device switch)."""
    runs = _mixed_shard_runs((0, 0))
    _assert_same_runs(runs[0], runs[1])


def test_mixed_c5_step_on_second_device_while_first_is_current():
    """BASELINE configs[4]'s per-GPU step (bench.MixedStep: quadrotors and MetaMaze3D frames on two HIP streams) built on cuda:1
    while cuda:0 stays the thread's current device — what rank r of `bench.py --gpus N --workload mixed` does when a launcher
    does not pin devices: states, rewards and every frame equal the same shard stepped on cuda:0."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    runs = _mixed_shard_runs((0, 1))
    _assert_same_runs(runs[0], runs[1])
