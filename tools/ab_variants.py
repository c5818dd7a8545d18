#!/usr/bin/env python3
"""A/B timing of diagnostic library builds in ONE process on ONE box (box-to-box variance is ~25 %):
python tools/ab_variants.py std xcd nt ...   (suffixes of libtwoarmy_hip_<suffix>.so, "std" = the shipped library;
AB_ENVS / AB_VARIANT select the batch size and the env variant, default 4096 / 6).
Each variant is loaded as its own ctypes handle; rounds are interleaved; all variants write into the SAME output
buffers (their HBM placement alone moves the time by up to 25 %) unless AB_SHARED_OUTPUTS=0."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
import twoarmy_amd  # noqa
from twoarmy_amd import _lib, engine as eng_mod  # noqa

T, N = 128, int(os.environ.get("AB_ENVS", "4096"))
VARIANT = int(os.environ.get("AB_VARIANT", "6"))
CODES = os.environ.get("AB_CODES", "0") == "1"          # uint8 code frames (TW_F_MATRIX_CODE)
VIEW = int(os.environ.get("AB_VIEW", "17"))
names = sys.argv[1:] or ["std"]
engines = []
base = _lib.LIB_PATH
for nm in names:
    _lib._lib = None
    _lib.LIB_PATH = base if nm == "std" else os.path.join(os.path.dirname(base), "libtwoarmy_hip_%s.so" % nm)
    lib = _lib.lib()                                    # binds signatures on this handle
    e = eng_mod.TwoarmyEngine(VARIANT, N, VIEW, seed=9981)
    shared = engines[0][4] if engines and os.environ.get("AB_SHARED_OUTPUTS", "1") == "1" else None
    engines.append((nm, lib, e, e.fill_actions(T), shared if shared is not None else e.alloc_outputs(T, matrix_codes=CODES)))
    o = engines[-1][4]
    print("%-8s obs %#x matrix %#x pos %#x reward %#x (matrix - obs = %d MiB + %d B)" % (
        nm, o["obs"].data_ptr(), o["matrix"].data_ptr(), o["pos"].data_ptr(), o["reward"].data_ptr(),
        (o["matrix"].data_ptr() - o["obs"].data_ptr()) >> 20, (o["matrix"].data_ptr() - o["obs"].data_ptr()) & 0xfffff), flush=True)
for rnd in range(4):
    for nm, lib, e, acts, out in engines:
        _lib._lib = lib
        ms = e.time_rollout(T, out, actions=acts, iters=20)
        print("round %d %-8s %.3f ms/launch  %.3f us/step" % (rnd, nm, ms, ms * 1e3 / T), flush=True)
