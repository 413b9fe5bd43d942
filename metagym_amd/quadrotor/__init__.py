"""Batched Quadrotor (mirrors metagym/quadrotor/__init__.py: id 'quadrotor-v0')."""
from .env import Quadrotor, DEFAULT_SIM_CONFIG

__all__ = ["Quadrotor", "DEFAULT_SIM_CONFIG"]
