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
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ.get("STAMP_LIB", "libtwoarmy_hip_stamp.so"))
from twoarmy_amd.engine import TwoarmyEngine  # noqa
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 6
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
eng = TwoarmyEngine(variant, N, 17, seed=9981)
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
print("v%d, %d envs: logic wave %.1f s_memtime ticks/step (%.0f per 128-step chunk); emit wave 1: %d tasks, poll %.0f cycles/task, work %.0f cycles/task" %
      (variant, N, a[:, 0].mean() / T, a[:, 0].mean(), a[:, 3].mean(), a[:, 1].mean() / a[:, 3].mean(), a[:, 2].mean() / a[:, 3].mean()))
print("wave 0: prologue %.0f ticks, logic -> kernel end %.0f ticks, whole kernel %.0f ticks" % (a[:, 4].mean(), a[:, 5].mean(), a[:, 6].mean()))
buf2 = (C.c_ulonglong * 3072)()
lib.tw_debug_stamps2.argtypes = [C.c_void_p]
assert lib.tw_debug_stamps2(buf2) == 0
b = np.array(buf2[:], dtype=np.float64).reshape(64, 16, 3)
print("prologue split (ticks): loads issued %.0f | loads landed + LDS fill + sync %.0f | verify %.0f | lane constants + logic state %.0f | staging + sync %.0f" % (b[:, 0, 0].mean(), b[:, 0, 1].mean(), b[:, 0, 2].mean(), b[:, 1, 0].mean(), b[:, 1, 1].mean()))
print("per-wave tasks (mean over 64 blocks):", np.round(b[:, :, 0].mean(0)).astype(int).tolist())
print("per-wave work cycles/task:", np.round(b[:, :, 2].mean(0) / np.maximum(b[:, :, 0].mean(0), 1)).astype(int).tolist())
print("per-wave poll cycles/task:", np.round(b[:, :, 1].mean(0) / np.maximum(b[:, :, 0].mean(0), 1)).astype(int).tolist())
buf3 = (C.c_ulonglong * 4096)()
lib.tw_debug_stamps3.argtypes = [C.c_void_p]
assert lib.tw_debug_stamps3(buf3) == 0
c = np.array(buf3[:], dtype=np.float64).reshape(64, 16, 4)
nb = min(64, (N + 15) // 16 if N >= 4096 else 64)
c = c[:nb]
t0 = c[:, :, 0].min()
print("kernel entry of the workgroups' wave 0, relative to the earliest wave of these %d workgroups: min %.0f median %.0f max %.0f"
      % (nb, (c[:, 0, 0] - t0).min(), np.median(c[:, 0, 0] - t0), (c[:, 0, 0] - t0).max()))
print("within a workgroup, entry of wave k relative to its wave 0 (median over workgroups):",
      np.round(np.median(c[:, :, 0] - c[:, :1, 0], axis=0)).astype(int).tolist())
print("entry -> loads issued per wave (median):", np.round(np.median(c[:, :, 1] - c[:, :, 0], axis=0)).astype(int).tolist())
print("loads issued -> past the first barrier per wave (median):", np.round(np.median(c[:, :, 2] - c[:, :, 1], axis=0)).astype(int).tolist())
