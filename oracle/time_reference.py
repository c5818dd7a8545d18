#!/usr/bin/env python3
"""Time the reference's own Python env (build container only; /root/reference never travels).
TEST INFRASTRUCTURE ONLY.  Results are quoted in BASELINE.md section 2 / DESIGN.md section 6."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import philox  # noqa: E402
import ref_harness as rh  # noqa: E402


def main(n_steps=10000):
    env_buffer, _ = rh.soa_modules()
    for variant in ("v6", "v4"):
        np.random.seed(9981)
        env = rh.make_env(variant)
        acts = philox.action_indices(9981, np.zeros(n_steps, np.uint32), np.arange(n_steps, dtype=np.uint32))
        acts = [6 if a == 4 else int(a) for a in acts]
        t0 = time.perf_counter()
        for a in acts:
            _, _, te, tr, _ = env.step(a)
            if te or tr:
                env.reset()
        dt = time.perf_counter() - t0
        et = env_buffer.Env_transact()
        t1 = time.perf_counter()
        for _ in range(2000):
            et.matrix_env(env)
        dm = (time.perf_counter() - t1) / 2000
        print("reference %s: env.step+reset %.0f steps/s (1 core, %d steps); matrix_env %.0f /s" %
              (variant, n_steps / dt, n_steps, 1.0 / dm), flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 10000)
