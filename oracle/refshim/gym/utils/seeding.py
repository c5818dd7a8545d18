"""gym.utils.seeding stand-in (see gym/__init__.py docstring)."""
import numpy as np

RandomNumberGenerator = np.random.Generator


def np_random(seed=None):
    ss = np.random.SeedSequence(seed)
    return np.random.Generator(np.random.PCG64(ss)), ss.entropy
