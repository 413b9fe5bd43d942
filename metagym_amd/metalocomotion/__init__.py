"""Batched MetaLocomotion walkers (mirrors metagym/metalocomotion/__init__.py: ids
meta-humanoid-v0, meta-ant-v0). Physics parity is UNPINNED (PyBullet is not in the reference tree);
see DESIGN.md §3.5."""
from . import variants
from .mjcf import Model, load_mjcf
from .walker_env import MetaHumanoidEnv, MetaAntEnv, WalkerBatchEnv

__all__ = ["MetaHumanoidEnv", "MetaAntEnv", "WalkerBatchEnv", "Model", "load_mjcf", "variants"]
