"""Shared helpers for the PPO tests (same deterministic weights as oracle/gen_golden.py::det_weights)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_cache = {}


def load_ppo_golden():
    if "ppo" not in _cache:
        _cache["ppo"] = dict(np.load(os.path.join(GOLDEN, "ppo.npz")))
    return _cache["ppo"]


def det_weights(module, seed):
    sd = {}
    for k, (name, prm) in enumerate(module.state_dict().items()):
        n = prm.numel()
        fan = max(1, n // prm.shape[0]) if prm.dim() > 1 else 1
        scale = (1.5 / np.sqrt(fan)) if prm.dim() > 1 else 0.05
        v = scale * np.sin(0.37 * np.arange(n, dtype=np.float64) + 1.7 * k + seed)
        sd[name] = torch.tensor(v.reshape(tuple(prm.shape)), dtype=prm.dtype)
    return sd
