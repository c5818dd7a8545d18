#!/usr/bin/env python3
"""Time tw_rollout with an alternative diagnostic build of the library: python tools/prof_variant_lib.py <suffix> [variant]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
import twoarmy_amd  # noqa
from twoarmy_amd import _lib
suffix = sys.argv[1]
if suffix != "std":
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libtwoarmy_hip_%s.so" % suffix)
from twoarmy_amd.engine import TwoarmyEngine  # noqa
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 6
eng = TwoarmyEngine(variant, 4096, 17, seed=9981)
T = 128
acts = eng.fill_actions(T)
out = eng.alloc_outputs(T)
for r in range(3):
    ms = eng.time_rollout(T, out, actions=acts, iters=10)
    print("%s v%d: %.3f ms/launch %.2f us/step" % (suffix, variant, ms, ms * 1e3 / T), flush=True)
