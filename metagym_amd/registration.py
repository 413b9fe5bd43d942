"""Id -> constructor table mirroring the reference's gym.register calls
(metagym/quadrotor/__init__.py:20-32, metagym/metamaze/__init__.py:21-54)."""
import importlib

registry = {}


def register(id, entry_point, kwargs=None):
    registry[id] = (entry_point, dict(kwargs or {}))


def make(id, **kwargs):
    if id not in registry:
        raise KeyError("unknown env id %r; known: %s" % (id, sorted(registry)))
    entry_point, defaults = registry[id]
    mod_name, cls_name = entry_point.split(":")
    cls = getattr(importlib.import_module(mod_name), cls_name)
    kw = dict(defaults)
    kw.update(kwargs)
    return cls(**kw)


register("quadrotor-v0", "metagym_amd.quadrotor:Quadrotor",
         kwargs={"dt": 0.01, "nt": 1000, "seed": 0, "task": "no_collision", "map_file": None,
                 "simulator_conf": None, "healthy_reward": 1.0})

# metagym/metamaze/__init__.py:21-54 (enable_render defaults to False here: there is no viewer)
register("meta-maze-continuous-3D-v0", "metagym_amd.metamaze:MetaMazeContinuous3D",
         kwargs={"enable_render": False, "render_scale": 480, "resolution": (256, 256), "max_steps": 5000,
                 "task_type": "SURVIVAL"})
register("meta-maze-discrete-3D-v0", "metagym_amd.metamaze:MetaMazeDiscrete3D",
         kwargs={"enable_render": False, "render_scale": 480, "resolution": (256, 256), "max_steps": 200,
                 "task_type": "SURVIVAL"})
register("meta-maze-2D-v0", "metagym_amd.metamaze:MetaMaze2D",
         kwargs={"enable_render": False, "max_steps": 200, "view_grid": 1, "task_type": "SURVIVAL"})

# metagym/metalocomotion/__init__.py:19-37
register("meta-humanoid-v0", "metagym_amd.metalocomotion:MetaHumanoidEnv",
         kwargs={"frame_skip": 4, "time_step": 0.005, "enable_render": False, "max_steps": 2000})
register("meta-ant-v0", "metagym_amd.metalocomotion:MetaAntEnv",
         kwargs={"frame_skip": 4, "time_step": 0.005, "enable_render": False, "max_steps": 2000})

# metagym/quadrupedal/__init__.py:6-21. The robot file is the caller's (`urdf=`: a1/a1.urdf ships with pybullet_data, not with
# the reference) and runs on this repo's articulated-body engine (metagym_amd.quadrupedal.A1Physics), or hand over a batched
# simulator of your own (`physics=`).
register("quadrupedal-v0", "metagym_amd.quadrupedal:A1GymEnv",
         kwargs={"action_limit": (0.75, 0.75, 0.75), "render": False, "on_rack": False, "random_dynamic": False, "ETG": 0, "ETG_T": 0.5,
                 "ETG_H": 20, "ETG_path": "", "task": "plane", "dynamic_param": {}})
