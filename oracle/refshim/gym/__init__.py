"""Inert stand-in for the `gym` package -- TEST INFRASTRUCTURE ONLY.

`gym` is not installed in the build container and cannot be fetched.  The
reference (/root/reference, pure Python) only needs gym for *base classes and
containers* (Env, Wrapper, spaces.*, seeding, register/make); none of the
arithmetic on the hot path (Twoarmy step, Grid.slice/rotate/encode, Env_transact,
Buffer_gridworld, PPO) lives in gym.  This shim provides those inert base
classes so that `oracle/gen_golden.py` can import the reference *in this
container* and record golden vectors.  It is authored from scratch, contains no
reference or gym code, is never imported by the product package, and never
travels with a claim of being gym.
"""
from . import spaces, core, utils            # noqa: F401
from .core import Env, Wrapper, ObservationWrapper  # noqa: F401
from .envs.registration import register, make      # noqa: F401

__version__ = "0.25.0-shim"
