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


def fill_gbs(t):
    base = t._base if t._base is not None else t
    base.fill_(1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        base.fill_(2)
    e1.record()
    torch.cuda.synchronize()
    return 10 * base.numel() * base.element_size() / (e0.elapsed_time(e1) * 1e-3) / 1e9


# is a "slow" buffer also slow for a plain linear fill?  (physical placement vs access pattern)
for k, o in enumerate(sets):
    print("set %d: rollout %.3f ms | plain fill of its obs buffer %.0f GB/s, matrix buffer %.0f GB/s"
          % (k, eng.time_rollout(T, o, actions=acts, iters=10), fill_gbs(o["obs"]), fill_gbs(o["matrix"])), flush=True)


# one slab per output set: obs, matrix and the scalar outputs carved out of ONE allocation
def slab_outputs(T, N, V=17):
    nb, ob, mb = V * V * 3, 880, 292 * 4
    sizes = [T * N * mb, T * N * ob, T * N * 8, T * N * 4, T * N, T * N]
    offs, o = [], 0
    for sz in sizes:
        offs.append(o)
        o += (sz + 4095) // 4096 * 4096
    slab = torch.empty(o, dtype=torch.uint8, device=eng.device)
    m = slab[offs[0]:offs[0] + sizes[0]].view(torch.float32).view(T, N, 292)[..., :289]
    ob_t = slab[offs[1]:offs[1] + sizes[1]].view(T, N, ob)[..., :nb].view(T, N, V, V, 3)
    return dict(obs=ob_t, matrix=m, pos=slab[offs[2]:offs[2] + sizes[2]].view(torch.float32).view(T, N, 2),
                reward=slab[offs[3]:offs[3] + sizes[3]].view(torch.float32).view(T, N),
                terminated=slab[offs[4]:offs[4] + sizes[4]].view(T, N), truncated=slab[offs[5]:offs[5] + sizes[5]].view(T, N))


slabs = [slab_outputs(T, N) for _ in range(K)]
for rnd in range(2):
    print("slab sets round %d: " % rnd + "  ".join("%.3f" % eng.time_rollout(T, o, actions=acts, iters=10) for o in slabs), flush=True)
print("separate sets again: " + "  ".join("%.3f" % eng.time_rollout(T, o, actions=acts, iters=10) for o in sets), flush=True)


# windows of ONE big allocation: are there fast and slow regions inside a single VA-contiguous slab?
del slabs
torch.cuda.empty_cache()
WIN = 1280 << 20
big = torch.empty(12 * WIN, dtype=torch.uint8, device=eng.device)


def window_outputs(base, T, N, V=17):
    nb, ob, mb = V * V * 3, 880, 292 * 4
    m = big[base:base + T * N * mb].view(torch.float32).view(T, N, 292)[..., :289]
    o0 = base + (T * N * mb + 4095) // 4096 * 4096
    ob_t = big[o0:o0 + T * N * ob].view(T, N, ob)[..., :nb].view(T, N, V, V, 3)
    return dict(sets[0], obs=ob_t, matrix=m)


wins = [window_outputs(k * WIN, T, N) for k in range(12)]
for rnd in range(2):
    print("12 windows of one 15 GiB slab, round %d: " % rnd +
          "  ".join("%.3f" % eng.time_rollout(T, o, actions=acts, iters=10) for o in wins), flush=True)
