"""gym.envs.registration stand-in (see gym/__init__.py docstring).

`make` instantiates the registered entry point with merged kwargs; real gym
0.25 with new_step_api=True only adds pass-through wrappers around it.
"""
import importlib

_REGISTRY = {}


def register(id, entry_point=None, kwargs=None, **_ignored):
    _REGISTRY[id] = (entry_point, dict(kwargs or {}))


def make(id, **kwargs):
    entry_point, base = _REGISTRY[id]
    kw = dict(base)
    kw.update(kwargs)
    kw.pop("new_step_api", None)
    if isinstance(entry_point, str):
        mod, _, attr = entry_point.partition(":")
        entry_point = getattr(importlib.import_module(mod), attr)
    return entry_point(**kw)
