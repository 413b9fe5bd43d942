"""metagym_amd — MI355X-native batched environment engine with the MetaGym env surface.

    import metagym_amd
    env = metagym_amd.make("quadrotor-v0", num_envs=65536, task="hovering_control")
    obs = env.reset()
    obs, reward, done, info = env.step(actions)      # torch-ROCm tensors, [num_envs, ...]

Everything is computed by hand-written gfx950 HIP kernels behind the C ABI in
include/metagym_hip.h; there is no CPU fallback.
"""
from .registration import make, register, registry  # noqa: F401

__version__ = "0.1.0"
