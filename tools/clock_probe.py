#!/usr/bin/env python3
"""Runs a 1 GB fill and the default rollout back to back (for rocprofv3 --pmc GRBM_GUI_ACTIVE: effective clock
= GRBM_GUI_ACTIVE / 8 / kernel time, MI355X_MICROARCH.md 'DVFS give-back')."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from twoarmy_amd.engine import TwoarmyEngine
x = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
eng = TwoarmyEngine(6, 4096, 17, seed=9981)
acts = eng.fill_actions(128)
out = eng.alloc_outputs(128)
for _ in range(30):
    x.fill_(1.0)
torch.cuda.synchronize()
for _ in range(30):
    eng.rollout(128, out, actions=acts)
torch.cuda.synchronize()
for _ in range(30):
    x.fill_(1.0)
torch.cuda.synchronize()
