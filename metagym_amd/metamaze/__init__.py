"""Batched MetaMaze (mirrors metagym/metamaze/__init__.py: ids meta-maze-2D-v0,
meta-maze-discrete-3D-v0, meta-maze-continuous-3D-v0)."""
from .maze_env import MetaMaze2D, MetaMazeDiscrete3D, MetaMazeContinuous3D
from .maze_task import MAZE_TASK_MANAGER, DeviceTaskTable, MazeTaskManager, MazeTaskSampler, TaskConfig

__all__ = ["MetaMaze2D", "MetaMazeDiscrete3D", "MetaMazeContinuous3D", "MazeTaskSampler", "MazeTaskManager",
           "MAZE_TASK_MANAGER", "TaskConfig", "DeviceTaskTable"]
