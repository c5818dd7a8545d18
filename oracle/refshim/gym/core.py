"""gym.core stand-in (see gym/__init__.py docstring)."""
from .utils import seeding


class Env:
    metadata = {}
    reward_range = (-float("inf"), float("inf"))
    action_space = None
    observation_space = None
    _np_random = None

    @property
    def np_random(self):
        if self._np_random is None:
            self._np_random, _ = seeding.np_random()
        return self._np_random

    @np_random.setter
    def np_random(self, value):
        self._np_random = value

    def reset(self, *, seed=None, return_info=False, options=None):
        if seed is not None:
            self._np_random, _ = seeding.np_random(seed)

    def step(self, action):
        raise NotImplementedError

    def close(self):
        pass

    @property
    def unwrapped(self):
        return self


class Wrapper(Env):
    def __init__(self, env, new_step_api=False):
        self.env = env

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.env, name)

    def step(self, action):
        return self.env.step(action)

    def reset(self, **kwargs):
        return self.env.reset(**kwargs)

    @property
    def unwrapped(self):
        return self.env.unwrapped


class ObservationWrapper(Wrapper):
    def reset(self, **kwargs):
        out = self.env.reset(**kwargs)
        if isinstance(out, tuple):
            return self.observation(out[0]), out[1]
        return self.observation(out)

    def step(self, action):
        out = self.env.step(action)
        return (self.observation(out[0]),) + tuple(out[1:])

    def observation(self, obs):
        raise NotImplementedError
