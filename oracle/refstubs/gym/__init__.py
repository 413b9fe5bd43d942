"""Minimal stand-in for OpenAI gym (old 4-tuple API) — test infrastructure, see ../README.md."""
from . import spaces, envs, utils, error  # noqa: F401
from .spaces import Space  # noqa: F401


class Env(object):
    action_space = None
    observation_space = None

    def reset(self):
        raise NotImplementedError

    def step(self, action):
        raise NotImplementedError

    def render(self, mode="human"):
        raise NotImplementedError

    def close(self):
        pass


def make(id, **kwargs):
    return envs.registration.make(id, **kwargs)


class Wrapper(Env):
    """gym.Wrapper (gym 0.18 core.py): forwards everything to `env` (quadrupedal/envs/env_wrappers/MonitorEnv.py)."""

    def __init__(self, env):
        self.env = env
        self.action_space = getattr(env, "action_space", None)
        self.observation_space = getattr(env, "observation_space", None)

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError("attempted to get missing private attribute '%s'" % name)
        return getattr(self.env, name)

    def step(self, action):
        return self.env.step(action)

    def reset(self, **kwargs):
        return self.env.reset(**kwargs)

    def close(self):
        return self.env.close()
