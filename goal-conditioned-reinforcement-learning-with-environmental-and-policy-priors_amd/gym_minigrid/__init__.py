"""Registry of the two Twoarmy ids (reference gym_minigrid/__init__.py:6-21).  `gym` is optional:
when importable the ids are also registered with it, otherwise `make()` below is the entry point."""
from .envs import Twoarmy_v4, Twoarmy_v6

_REGISTRY = {
    "MiniGrid-twoarmy-17x17-v4": (Twoarmy_v4, {"size": 17}),
    "MiniGrid-twoarmy-17x17-v6": (Twoarmy_v6, {"size": 17}),
}


def register_minigrid_envs():
    try:
        from gym.envs.registration import register
    except Exception:
        return False
    for env_id, (cls, kw) in _REGISTRY.items():
        try:
            register(id=env_id, entry_point=cls, kwargs=kw)
        except Exception:
            pass
    return True


def make(id, **kwargs):
    cls, base = _REGISTRY[id]
    kw = dict(base)
    kw.update(kwargs)
    kw.pop("new_step_api", None)
    return cls(**kw)
