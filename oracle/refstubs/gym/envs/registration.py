import importlib

_REGISTRY = {}


def register(id, entry_point=None, kwargs=None, **_):
    _REGISTRY[id] = (entry_point, dict(kwargs or {}))


def make(id, **kwargs):
    entry_point, defaults = _REGISTRY[id]
    mod_name, cls_name = entry_point.split(":")
    cls = getattr(importlib.import_module(mod_name), cls_name)
    kw = dict(defaults)
    kw.update(kwargs)
    return cls(**kw)
