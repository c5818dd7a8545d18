#!/usr/bin/env python3
"""Throughput of the general MiniGrid view kernel (mg_gen_obs) and base step (mg_step) on random worlds.
Prints one JSON line per configuration: views/s and the HBM roofline fraction for the algorithmic bytes
(window cells of three planes read + image and mask written).

  python tools/view_bench.py [--envs 262144] [--size 17] [--iters 20]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from twoarmy_amd import minigrid_view as mv  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=262144)
ap.add_argument("--size", type=int, default=17)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda", 0)
N, W = a.envs, a.size
g = torch.Generator(device="cpu").manual_seed(1)
ty = torch.tensor([1, 1, 1, 1, 1, 2, 2, 4, 5, 6, 8], dtype=torch.uint8)[torch.randint(0, 11, (N, W * W), generator=g)].to(dev)
co = torch.randint(0, 6, (N, W * W), generator=g, dtype=torch.uint8).to(dev)
st = torch.where(ty == 4, torch.randint(0, 3, (N, W * W), generator=g, dtype=torch.uint8).to(dev), torch.zeros_like(ty))
ax = torch.randint(0, W, (N,), generator=g, dtype=torch.int32).to(dev)
ay = torch.randint(0, W, (N,), generator=g, dtype=torch.int32).to(dev)
d = torch.randint(0, 4, (N,), generator=g, dtype=torch.int32).to(dev)


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters * 1e-3


for V, see in ((7, False), (7, True), (17, False), (17, True)):
    out = torch.empty((N, V, V, 3), dtype=torch.uint8, device=dev)
    s = timed(lambda: mv.gen_obs(ty, co, st, W, W, ax, ay, d, V, see, None, want_mask=False, out=out))
    nbytes = N * (3 * V * V + 3 * V * V + 12)
    print(json.dumps({"kernel": "mg_gen_obs", "envs": N, "world": "%dx%d" % (W, W), "view": V, "see_through_walls": see,
                      "ms": s * 1e3, "views_per_s": N / s, "algorithmic_GBs": nbytes / s / 1e9,
                      "frac_of_8TBs": nbytes / s / 8e12}), flush=True)
sc = torch.zeros(N, dtype=torch.int32, device=dev)
act = torch.randint(0, 4, (N,), generator=g, dtype=torch.int32).to(dev)
s = timed(lambda: mv.step(ty, st, W, W, act, ax, ay, d, sc, 1 << 30))
print(json.dumps({"kernel": "mg_step", "envs": N, "ms": s * 1e3, "steps_per_s": N / s}), flush=True)
