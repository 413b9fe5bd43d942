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
