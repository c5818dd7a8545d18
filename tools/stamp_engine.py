#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of tw_rollout_kernel from the -DTW_STAMP build (make -C csrc stamp).
Shares only -- the stamped build's run time is not a performance number."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import twoarmy_amd  # noqa: E402
from twoarmy_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libtwoarmy_hip_stamp.so")
from twoarmy_amd.engine import TwoarmyEngine  # noqa: E402

variant = int(sys.argv[1]) if len(sys.argv) > 1 else 6
E = int(sys.argv[2]) if len(sys.argv) > 2 else 0
eng = TwoarmyEngine(variant, 4096, 17, seed=9981)
eng.set_envs_per_wave(E)
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
import numpy as np  # noqa: E402
a = np.array(buf[:], dtype=np.float64).reshape(64, 8)
names = ["loop-top/actions", "pre-part1", "part1 (logic)", "select+gathers", "part2 (logic+scalars out)",
         "pack+store", "autoreset", "-"]
tot = a[:, :7].sum(1).mean()
print("variant v%d E=%d: %.0f cycles per wave-step (stamped build)" % (variant, E, tot / T))
for i in range(7):
    print("  %-28s %7.0f cycles/step  %5.1f %%" % (names[i], a[:, i].mean() / T, 100 * a[:, i].mean() / tot))
