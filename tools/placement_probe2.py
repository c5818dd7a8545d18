#!/usr/bin/env python3
"""How the physical backing of the output slab changes the rollout time (VERDICT r1 item 5 / DESIGN 9.1).
Configurations: torch caching allocator; engine slab via hipMalloc; via separately created hipMemCreate chunks of
2 MiB ... 256 MiB mapped into one virtual range (creation order / reverse / shuffled); one chunk for the slab; and a
contiguous slab with the obs stream moved by a gap.  K sets per configuration (kept alive), the same 4096 x 128
rollout timed into each in interleaved rounds."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from twoarmy_amd.engine import TwoarmyEngine  # noqa

T, N, K = 128, 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 3
which = sys.argv[2] if len(sys.argv) > 2 else "all"
eng = TwoarmyEngine(6, N, 17, seed=9981)
acts = eng.fill_actions(T)
CONFIGS = [("torch", None, 0, 0), ("hipMalloc", 0, 0, 0),
           ("c2M", 1, 0, 0), ("c4M", 2, 0, 0), ("c8M", 3, 0, 0), ("c16M", 4, 0, 0), ("c32M", 5, 0, 0), ("c64M", 6, 0, 0),
           ("c256M", 8, 0, 0), ("one", 99, 0, 0),
           ("c2M-rev", 1, 1, 0), ("c2M-shuf", 1, 2, 0), ("c32M-shuf", 5, 2, 0),
           ("hipMalloc+gap1M", 0, 0, 1024), ("hipMalloc+gap33M", 0, 0, 33 * 1024), ("hipMalloc+gap777K", 0, 0, 777)]
if which != "all":
    CONFIGS = [c for c in CONFIGS if c[0] in which.split(",")]
sets = []
for name, backing, order, gap in CONFIGS:
    for k in range(K):
        if backing is None:
            o = eng.alloc_outputs(T, slab=False)
        else:
            os.environ["TW_SLAB_BACKING"] = str(backing)
            os.environ["TW_SLAB_ORDER"] = str(order)
            os.environ["TW_SLAB_GAP_KB"] = str(gap)
            o = eng.alloc_outputs(T)
        sets.append((name, k, o))
print("allocated %d sets, %.1f GB" % (len(sets), len(sets) * 1.09), flush=True)
res = {}
for rnd in range(3):
    for name, k, o in sets:
        ms = eng.time_rollout(T, o, actions=acts, iters=6)
        res.setdefault(name, {}).setdefault(k, []).append(ms)
    print("round %d done" % rnd, flush=True)
summary = {}
for name, per in res.items():
    best = [round(min(v[1:]), 4) for v in per.values()]
    summary[name] = best
    print("%-18s %s" % (name, "  ".join("%.4f" % b for b in best)), flush=True)
print(json.dumps({"T": T, "N": N, "ms_per_launch": summary}))
