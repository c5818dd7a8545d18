"""Harness that imports the *reference* (read-only, /root/reference) in the build container.

TEST INFRASTRUCTURE ONLY.  Used by oracle/gen_golden.py to record golden vectors and by
oracle/time_reference.py to time the reference's CPU env.  `/root/reference` does not exist on
the GPU box, so nothing under tests/ (-m gpu), bench.py or the product package imports this.

What it does:
  * puts oracle/refshim (inert `gym` base classes + side-effect-only stubs for turtle,
    tensorboardX, torchvision, seaborn) and the reference roots on sys.path;
  * forces the Agg matplotlib backend;
  * neutralises the reference's heatmap() (it writes to an absolute /home/... path);
  * offers `patched_choice(recorder)` to replace np.random.choice inside the reference's
    Twoarmy modules with a supplied draw source, so the data-dependent global MT19937 stream
    (twoarmy_v4.py:117,149,184,190,215,221,303,310) can be replayed slot-by-slot.
"""
import contextlib
import os
import sys

REF_ROOT = os.environ.get("TWOARMY_REFERENCE_ROOT", "/root/reference")
_SHIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "refshim")


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "gym_minigrid"))


def setup():
    if not available():
        raise RuntimeError("reference not present at %s (expected on the GPU box)" % REF_ROOT)
    os.environ.setdefault("MPLBACKEND", "Agg")
    for p in (os.path.join(REF_ROOT, "soa"), REF_ROOT, _SHIM):
        if p not in sys.path:
            sys.path.insert(0, p)
    import gym_minigrid  # noqa: F401  (reference package)
    gym_minigrid.register_minigrid_envs()
    return gym_minigrid


def make_env(variant="v6", **kw):
    """Instantiate the reference Twoarmy env the way soa/train_ppo.py:80-85 does."""
    setup()
    import gym
    kw.setdefault("tile_size", 17)
    return gym.make("MiniGrid-twoarmy-17x17-%s" % variant, new_step_api=True, **kw)


def soa_modules():
    """Import soa.env_buffer / soa.agent.PPO with heatmap neutralised (absolute output path)."""
    setup()
    import env_buffer
    from agent import PPO as ppo_mod
    ppo_mod.heatmap = lambda *a, **k: None
    return env_buffer, ppo_mod


class SlotRecorder:
    """Callable replacing np.random.choice(range(a,b),1): returns draw_fn(lo, n) and logs it."""

    def __init__(self, draw_fn):
        self.draw_fn = draw_fn
        self.log = []

    def __call__(self, rng, size=None, replace=True, p=None):
        import numpy as np
        lo, n = rng[0], len(rng)
        v = int(self.draw_fn(lo, n))
        assert lo <= v < lo + n
        self.log.append((lo, n, v))
        return np.array([v])


@contextlib.contextmanager
def patched_choice(recorder):
    """Patch `np.random.choice` as seen by the reference's twoarmy modules only."""
    setup()
    import numpy as np
    from gym_minigrid.envs import twoarmy_v4, twoarmy_v6

    class _RandomProxy:
        def __init__(self, real, choice):
            self._real = real
            self.choice = choice

        def __getattr__(self, name):
            return getattr(self._real, name)

    class _NpProxy:
        def __init__(self, real, choice):
            self._real = real
            self.random = _RandomProxy(real.random, choice)

        def __getattr__(self, name):
            return getattr(self._real, name)

    saved = (twoarmy_v4.np, twoarmy_v6.np)
    proxy = _NpProxy(np, recorder)
    twoarmy_v4.np = proxy
    twoarmy_v6.np = proxy
    try:
        yield recorder
    finally:
        twoarmy_v4.np, twoarmy_v6.np = saved
