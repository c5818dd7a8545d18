#!/usr/bin/env python3
"""Interleaved record layout probe: obs and matrix of one env-step in ONE 2048-byte record (1168 B matrix row + 880 B
obs row) instead of two separate streams.  Uses the existing pitch arguments of tw_rollout, no kernel change."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from twoarmy_amd.engine import TwoarmyEngine  # noqa

T, N, K = 128, 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 4
eng = TwoarmyEngine(6, N, 17, seed=9981)
acts = eng.fill_actions(T)


def interleaved(mat_first=True):
    buf = torch.empty((T, N, 2048), dtype=torch.uint8, device=eng.device)
    mo, oo = (0, 1168) if mat_first else (880, 0)
    m = buf[..., mo:mo + 1168].view(torch.float32)[..., :289]
    o = buf[..., oo:oo + 867].view(T, N, 17, 17, 3)
    base = eng.alloc_outputs(T, slab=False, obs=False, matrix=False)
    return dict(base, obs=o, matrix=m), buf


ref = eng.alloc_outputs(T, slab=False)
state = eng.get_state()
eng.rollout(T, ref, actions=acts)
sets = [("separate(torch)", ref, None)]
for k in range(K):
    o, b = interleaved(True)
    sets.append(("interleaved mat|obs [%d]" % k, o, b))
for k in range(2):
    o, b = interleaved(False)
    sets.append(("interleaved obs|mat [%d]" % k, o, b))
for k in range(K):
    sets.append(("separate(torch) [%d]" % k, eng.alloc_outputs(T, slab=False), None))
# correctness of the interleaved views against the separate layout (same state, same actions)
eng.set_state(*state)
eng.rollout(T, sets[1][1], actions=acts)
torch.cuda.synchronize()
for k in ("obs", "matrix", "pos", "reward", "terminated", "truncated"):
    assert torch.equal(sets[1][1][k], ref[k]), k
print("interleaved layout == separate layout (bit-exact)", flush=True)
res = {}
for rnd in range(3):
    for name, o, _ in sets:
        res.setdefault(name, []).append(eng.time_rollout(T, o, actions=acts, iters=8))
for name, v in res.items():
    print("%-28s %s" % (name, "  ".join("%.4f" % x for x in v)), flush=True)
print(json.dumps({k: round(min(v[1:]), 4) for k, v in res.items()}))
