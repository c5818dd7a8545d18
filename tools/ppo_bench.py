#!/usr/bin/env python3
"""Full-PPO timing (BASELINE configs[2]): N envs x T-step rollouts with the actor in the loop, then the PPO
update (K epochs, minibatches of M) on the collected samples.  Prints one JSON line; not the headline metric
(bench.py is), kept for the policy-side numbers in DESIGN.md section 6.

  python tools/ppo_bench.py [--envs 4096] [--T 128] [--minibatch 32768] [--k_epochs 1] [--updates 2]
                            [--amp fp32|bf16] [--frame_codes] [--her] [--variant 6]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--nhwc" in sys.argv:
    os.environ["PYTORCH_MIOPEN_SUGGEST_NHWC"] = "1"          # must be set before torch is imported
import torch  # noqa: E402
from twoarmy_amd.engine import TwoarmyEngine  # noqa: E402
from twoarmy_amd.soa.agent.PPO import PPO  # noqa: E402
from twoarmy_amd.soa.ppo_vec import VecPPOTrainer  # noqa: E402

FWD_FLOP_PER_SAMPLE_PER_NET = 47.5e6

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--T", type=int, default=128)
ap.add_argument("--minibatch", type=int, default=32768)
ap.add_argument("--k_epochs", type=int, default=1)
ap.add_argument("--updates", type=int, default=2)
ap.add_argument("--variant", type=int, default=6)
ap.add_argument("--amp", default="fp32", choices=["fp32", "bf16"])
ap.add_argument("--frame_codes", action="store_true")
ap.add_argument("--her", action="store_true")
ap.add_argument("--unfused", action="store_true", help="with --nhwc: plain Conv2d + ReLU modules (A/B of the fused epilogues)")
ap.add_argument("--no_value_reuse", action="store_true")
ap.add_argument("--nhwc", action="store_true", help="channels-last conv stack (PYTORCH_MIOPEN_SUGGEST_NHWC=1)")
a = ap.parse_args()


def _heartbeat():                              # MIOpen's first-call kernel search can take minutes per shape
    import threading
    t0 = time.perf_counter()

    def beat():
        while True:
            time.sleep(60)
            print("... %.0f s" % (time.perf_counter() - t0), file=sys.stderr, flush=True)
    threading.Thread(target=beat, daemon=True).start()


_heartbeat()
torch.manual_seed(9981)
eng = TwoarmyEngine(a.variant, a.envs, 17, seed=9981)
agent = PPO()
agent.K_epochs = a.k_epochs
agent.amp_dtype = torch.bfloat16 if a.amp == "bf16" else None
if a.nhwc:
    agent.to(eng.device)
    agent.use_nhwc()
    if a.unfused:
        from twoarmy_amd.soa.agent.net import all_net
        all_net.TINet._convs = lambda self, img: self.cnn_base(img)
tr = VecPPOTrainer(agent, eng, rollout_steps=a.T, minibatch=a.minibatch, frame_codes=a.frame_codes)
tr.reuse_next_values = not a.no_value_reuse
tr.time_phases = True
roll, upd, recs = [], [], []
for u in range(a.updates + 1):                 # first pass = warm-up (MIOpen find, allocator)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.collect()
    if a.her:
        tr.relabel()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    n_her = 0 if tr.her is None else int(tr.her["t"].numel())
    la, lv = tr.update()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    tr.carry_over()
    if u:
        roll.append(t1 - t0)
        upd.append(t2 - t1)
        recs.append(n_her)
    print("pass %d rollout %.3fs update %.3fs her %d losses %.5f %.5f" % (u, t1 - t0, t2 - t1, n_her, float(la), float(lv)),
          file=sys.stderr, flush=True)
S = a.envs * a.T
r, w = sum(roll) / len(roll), sum(upd) / len(upd)
samples = S + sum(recs) / len(recs)
flop_roll = S * FWD_FLOP_PER_SAMPLE_PER_NET                                   # actor forward per env-step
flop_upd = samples * FWD_FLOP_PER_SAMPLE_PER_NET * (2 + a.k_epochs * 3 * 2)    # 2 critic fwd (targets) + K x (fwd+bwd) x 2 nets
print(json.dumps({"workload": "full PPO, twoarmy-v%d, %d envs x %d steps, minibatch %d, K=%d, GEMM dtype %s%s%s"
                              % (a.variant, a.envs, a.T, a.minibatch, a.k_epochs, a.amp,
                                 ", code frames" if a.frame_codes else "", ", HER" if a.her else ""),
                  "rollout_s": r, "update_s": w, "env_steps_per_s_rollout": S / r, "env_steps_per_s_loop": S / (r + w),
                  "rollout_TFLOPs": flop_roll / r / 1e12, "update_TFLOPs": flop_upd / w / 1e12,
                  "update_targets_s": tr.last_update_timing["targets_s"], "update_epoch_s": tr.last_update_timing["epoch_s"],
                  "nhwc": a.nhwc, "miopen_find_mode": os.environ.get("MIOPEN_FIND_MODE", "default"),
                  "her_records": sum(recs) / len(recs), "peak_mem_GB": torch.cuda.max_memory_allocated() / 1e9}))
eng.close()
