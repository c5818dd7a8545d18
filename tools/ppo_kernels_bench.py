#!/usr/bin/env python3
"""Achieved HBM bandwidth of the PPO-side HIP kernels (csrc/ppo_kernels.hip) at BASELINE configs[2] sizes
(T = 128 steps x N = 4096 envs = 524 288 samples; minibatch 32 768).  One JSON line per kernel:
algorithmic bytes / HIP-event time, as a fraction of the 8 TB/s peak."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from twoarmy_amd import ppo_ops  # noqa: E402

dev = torch.device("cuda", 0)
T, N, B, A = 128, 4096, 32768, 5
g = torch.Generator(device="cpu").manual_seed(0)


def timed(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def report(name, nbytes, s, note=""):
    print(json.dumps({"kernel": name, "us": s * 1e6, "algorithmic_MB": nbytes / 1e6, "GBs": nbytes / s / 1e9,
                      "frac_of_8TBs": nbytes / s / 8e12, "note": note}), flush=True)


probs = torch.softmax(torch.randn(T * N, A, generator=g), 1).to(dev)
u = torch.rand(T * N, generator=g).to(dev)
report("ppo_sample (524288 x 5)", T * N * (A * 4 + 4 + 4 + 4), timed(lambda: ppo_ops.sample(probs, u)))
report("ppo_sample, Philox uniforms", T * N * (A * 4 + 4 + 4), timed(lambda: ppo_ops.sample(probs, None, seed=1, offset=0)))

r, v, nv = (torch.randn(T, N, generator=g).to(dev) for _ in range(3))
done = (torch.rand(T, N, generator=g) < 0.02).to(torch.uint8).to(dev)
report("ppo_gae lambda=0 (reference targets)", T * N * (3 * 4 + 3 * 4),
       timed(lambda: ppo_ops.gae(r, v, nv, None, gamma=0.99, lam=0.0, use_done_mask=False)))
report("ppo_gae lambda=0.95 + done mask", T * N * (3 * 4 + 1 + 3 * 4),
       timed(lambda: ppo_ops.gae(r, v, nv, done, gamma=0.99, lam=0.95, use_done_mask=True)))
adv = torch.randn(T * N, generator=g).to(dev)
report("ppo_adv_norm (2 passes)", T * N * (4 + 4 + 4), timed(lambda: ppo_ops.adv_norm_(adv)))

pb = torch.softmax(torch.randn(B, A, generator=g), 1).to(dev).requires_grad_(True)
val = torch.randn(B, 1, generator=g).to(dev).requires_grad_(True)
act = torch.randint(0, A, (B,), generator=g, dtype=torch.int32).to(dev)
olp, ad, tg = (torch.randn(B, 1, generator=g).to(dev) for _ in range(3))
report("ppo_loss_fwd_bwd (minibatch 32768)", B * (A * 4 + 4 + 4 + 4 + 4 + 4 + A * 4 + 4),
       timed(lambda: ppo_ops.ppo_losses(pb, val, act, olp, ad, tg, clip=0.1, ent_coef=0.01)), "launch-bound at this size")

frames = torch.randn(T + 4, N, 292, generator=g).to(dev)[..., :289]
codes = torch.randint(0, 4, (T + 4, N, 304), generator=g, dtype=torch.uint8).to(dev)[..., :289]
pos = torch.randn(T + 4, N, 2, generator=g).to(dev)
k = torch.randint(3, T + 3, (B,), generator=g, dtype=torch.int32).to(dev)
n = torch.randint(0, N, (B,), generator=g, dtype=torch.int32).to(dev)
age = torch.randint(0, 50, (B,), generator=g, dtype=torch.int32).to(dev)
init_f, init_p = torch.randn(289, generator=g).to(dev), torch.randn(2, generator=g).to(dev)
report("ppo_gather_stack f32 frames (32768 x 4 x 289)", B * 4 * 289 * 8,
       timed(lambda: ppo_ops.gather_stack(frames, pos, k, n, age, init_f, init_p)))
report("ppo_gather_stack_u8 code frames", B * 4 * 289 * 5,
       timed(lambda: ppo_ops.gather_stack(codes, pos, k, n, age, init_f, init_p)))
term = (torch.rand(T, N, generator=g) < 0.01).to(torch.uint8).to(dev)
trunc = (torch.rand(T, N, generator=g) < 0.02).to(torch.uint8).to(dev)
age0 = torch.zeros(N, dtype=torch.int32, device=dev)
report("ppo_age_scan", T * N * (1 + 1 + 4), timed(lambda: ppo_ops.age_scan(term, trunc, age0)))
p2 = torch.randint(1, 16, (T, N, 2), generator=g).float().to(dev)
rw = torch.randn(T, N, generator=g).to(dev)
report("ppo_her_relabel (count + scan + emit, one host sync)", T * N * (8 + 1 + 1 + 4) * 2,
       timed(lambda: ppo_ops.her_relabel(p2, term, trunc, age0, rw, seed=1), iters=5), "includes the record-count sync")
