"""`BulletClient` of pybullet_utils: one physics world per client (env_bases.py:36-39). TEST INFRASTRUCTURE."""
import pybullet


class BulletClient(object):
    def __init__(self, connection_mode=None):
        self._world = pybullet.World()

    def __getattr__(self, name):
        return getattr(self._world, name)
