#!/usr/bin/env python3
"""Does the HBM placement of the output buffers change the rollout time?  One engine, K output sets allocated one
after the other (all kept alive), each timed in interleaved rounds."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from twoarmy_amd.engine import TwoarmyEngine  # noqa

T, N, K = 128, 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 8
eng = TwoarmyEngine(6, N, 17, seed=9981)
acts = eng.fill_actions(T)
sets = [eng.alloc_outputs(T) for _ in range(K)]
for k, o in enumerate(sets):
    print("set %d obs %#x matrix %#x pos %#x reward %#x term %#x" % (k, o["obs"].data_ptr(), o["matrix"].data_ptr(),
                                                                 o["pos"].data_ptr(), o["reward"].data_ptr(),
                                                                 o["terminated"].data_ptr()), flush=True)
for rnd in range(3):
    print("round %d: " % rnd + "  ".join("%.3f" % eng.time_rollout(T, o, actions=acts, iters=10) for o in sets), flush=True)
# mix: big streams of one set, scalars of another
a, b = sets[0], sets[-1]
mix = dict(a, pos=b["pos"], reward=b["reward"], terminated=b["terminated"], truncated=b["truncated"])
print("set0 streams + last set scalars: %.3f" % eng.time_rollout(T, mix, actions=acts, iters=10))
mix = dict(b, pos=a["pos"], reward=a["reward"], terminated=a["terminated"], truncated=a["truncated"])
print("last set streams + set0 scalars: %.3f" % eng.time_rollout(T, mix, actions=acts, iters=10))
mix = dict(a, matrix=b["matrix"])
print("set0 obs + last matrix: %.3f" % eng.time_rollout(T, mix, actions=acts, iters=10))
