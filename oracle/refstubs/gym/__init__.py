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
