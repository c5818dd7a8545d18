"""MI355X-native MiniGrid-Twoarmy step/observation engine + PPO rollout/update path.

Drop-in for the hot path of widkyyu/Goal-conditioned-Reinforcement-Learning-with-environmental-and-policy-priors
(gym_minigrid Twoarmy envs, soa/env_buffer.py, soa/agent/PPO.py, soa/train_ppo.py).  All env compute is
hand-written HIP for gfx950 behind the C ABI in include/twoarmy.h; see DESIGN.md.
"""
from . import _lib  # noqa: F401

__version__ = "0.1.0"


def build(force=False):
    return _lib.build(force=force)
