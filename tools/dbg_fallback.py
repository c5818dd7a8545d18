#!/usr/bin/env python3
"""Diagnostic: pipelined vs sequential rollout, where they differ and how many workgroups fell back."""
import gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from twoarmy_amd.engine import TwoarmyEngine

def once(variant, N, T):
    a, b = TwoarmyEngine(variant, N, 17, seed=9981), TwoarmyEngine(variant, N, 17, seed=9981)
    b.set_pipeline(False)
    acts = a.fill_actions(T)
    oa, ob = a.alloc_outputs(T), b.alloc_outputs(T)
    for o in (oa, ob):
        o["obs"].fill_(0x77)
    a.rollout(T, oa, actions=acts); b.rollout(T, ob, actions=acts)
    torch.cuda.synchronize()
    msg = []
    for k in oa:
        if not torch.equal(oa[k], ob[k]):
            d = (oa[k] != ob[k])
            while d.dim() > 2:
                d = d.any(-1)
            tn = torch.nonzero(d)
            msg.append("%s: %d (t,n) differ, first %s last %s, envs %s" % (k, tn.shape[0], tn[0].tolist(), tn[-1].tolist(), sorted(set(tn[:, 1].tolist()))[:12]))
    unw_a = int((oa["obs"].flatten(2) == 0x77).all(-1).sum()); unw_b = int((ob["obs"].flatten(2) == 0x77).all(-1).sum())
    print("v%d N=%d T=%d fallback wgs %d unwritten obs rows pipelined %d sequential %d | %s" % (
        variant, N, T, a.fallback_count(), unw_a, unw_b, "; ".join(msg) or "identical"), flush=True)

for rep in range(2):
    for variant in (6, 4):
        once(variant, 777, 150)
        gc.collect()
once(4, 130, 40); once(4, 777, 150); once(6, 96, 40); once(4, 777, 150)
