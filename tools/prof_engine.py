#!/usr/bin/env python3
"""Ablation timing of tw_rollout_kernel on the GPU box (HIP-event kernel time, one process).
usage: python tools/prof_engine.py [--envs 4096] [--T 128] [--variant 6] [--view 17]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from twoarmy_amd.engine import TwoarmyEngine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--T", type=int, default=128)
ap.add_argument("--variant", type=int, default=6)
ap.add_argument("--view", type=int, default=17)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
eng = TwoarmyEngine(a.variant, a.envs, a.view, seed=9981)
acts = eng.fill_actions(a.T)
full = eng.alloc_outputs(a.T)
variants = {
    "full": full,
    "no_obs": dict(full, obs=None),
    "no_matrix": dict(full, matrix=None),
    "scalars_only": dict(full, obs=None, matrix=None),
    "nothing": dict(obs=None, matrix=None, pos=None, reward=None, terminated=None, truncated=None),
}
for rnd in range(3):
    for name, out in variants.items():
        for use_acts in (True, False):
            ms = eng.time_rollout(a.T, out, actions=acts if use_acts else None, iters=a.iters)
            nb = 0
            for k, v in out.items():
                if v is not None:
                    nb += v.numel() * v.element_size()
            print("round %d %-13s actions=%-6s %8.3f ms/launch  %7.2f us/step  %7.1f GB/s written  %6.0f M env-steps/s"
                  % (rnd, name, "hbm" if use_acts else "philox", ms, ms * 1e3 / a.T, nb / ms / 1e6,
                     a.envs * a.T / ms / 1e3), flush=True)
