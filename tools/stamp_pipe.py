#!/usr/bin/env python3
"""Diagnostic (-DTW_STAMP build): where the pipelined kernel's waves spend their cycles."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa
import torch  # noqa
import twoarmy_amd  # noqa
from twoarmy_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libtwoarmy_hip_stamp.so")
from twoarmy_amd.engine import TwoarmyEngine  # noqa
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 6
eng = TwoarmyEngine(variant, 4096, 17, seed=9981)
T = 128
acts = eng.fill_actions(T)
out = eng.alloc_outputs(T)
for _ in range(3):
    eng.rollout(T, out, actions=acts)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 512)()
lib = _lib.lib()
lib.tw_debug_stamps.argtypes = [C.c_void_p]
assert lib.tw_debug_stamps(buf) == 0
a = np.array(buf[:], dtype=np.float64).reshape(64, 8)
print("v%d: logic wave %.0f cycles/step; emit wave 1: %d tasks, poll %.0f cycles/task, work %.0f cycles/task" %
      (variant, a[:, 0].mean() / T, a[:, 3].mean(), a[:, 1].mean() / a[:, 3].mean(), a[:, 2].mean() / a[:, 3].mean()))
