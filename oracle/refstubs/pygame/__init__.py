"""Headless stand-in for pygame (image IO through PIL) — test infrastructure, see ../README.md."""
import numpy as np
from . import font  # noqa: F401

K_LEFT, K_RIGHT, K_UP, K_DOWN = 0, 1, 2, 3


def init():
    return (0, 0)


class Surface(object):
    def __init__(self, size=(0, 0), array=None):
        self._array = array  # (W, H, 3) uint8, pygame surfarray convention
        self._size = size

    def get_width(self):
        return self._size[0]

    def get_height(self):
        return self._size[1]

    def fill(self, *_a, **_k):
        pass

    def blit(self, *_a, **_k):
        pass


def Color(*a):
    return a


class _Image(object):
    @staticmethod
    def load(path):
        from PIL import Image
        im = Image.open(path).convert("RGB")
        arr = np.transpose(np.asarray(im, dtype=np.uint8), (1, 0, 2)).copy()  # (H,W,3)->(W,H,3)
        return Surface(size=im.size, array=arr)

    @staticmethod
    def save(*_a, **_k):
        pass


class _Surfarray(object):
    @staticmethod
    def array3d(surface):
        return np.array(surface._array, dtype=np.uint8)

    @staticmethod
    def make_surface(arr):
        return Surface(size=arr.shape[:2], array=None)


class _Key(object):
    @staticmethod
    def get_pressed():
        return [0] * 512


class _Time(object):
    @staticmethod
    def delay(_ms):
        pass


image = _Image()
surfarray = _Surfarray()
key = _Key()
time = _Time()
