"""Diagnostic: the 8 logged losses of PPO.update on the golden buffer -- literal NCHW modules vs the channels-last fused
path vs the reference's log, and per-step gradient differences between the two paths at identical weights."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import twoarmy_amd  # noqa
from test_ppo_common import det_weights, load_ppo_golden  # noqa
from twoarmy_amd.soa.agent.PPO import PPO  # noqa

DEV = "cuda:0"
g = load_ppo_golden()
buf = {k: g["buf_" + k] for k in ("s", "a", "p", "g", "r", "d", "a_logp")}


def run(fused, unfused_epilogue=False):
    agent = PPO()
    agent.actor.load_state_dict(det_weights(agent.actor, 1))
    agent.critic.load_state_dict(det_weights(agent.critic, 2))
    agent.batch_size, agent.K_epochs = 16, 2
    agent.to(DEV)
    if fused:
        agent.use_nhwc()
    agent.update(buf, DEV, 0, permutations=g["upd_perms"])
    sc = agent.writer.scalars
    return (np.array([v for _, v in sc["loss/action_loss_update"]]), np.array([v for _, v in sc["loss/value_loss_update"]]))


for name, fused, det in (("literal", False, False), ("fused", True, False), ("literal+det", False, True), ("fused+det", True, True)):
    torch.backends.cudnn.deterministic = det
    la, lv = run(fused)
    print("%-8s action |d| %s" % (name, np.array2string(np.abs(la - g["upd_action_loss"]), precision=1)))
    print("%-8s value  |d| %s" % (name, np.array2string(np.abs(lv - g["upd_value_loss"]), precision=1)))

# gradients at identical weights: literal vs fused, first minibatch
from twoarmy_amd import ppo_ops  # noqa
grads = {}
torch.backends.cudnn.deterministic = "--det" in sys.argv
print("gradients with cudnn.deterministic =", torch.backends.cudnn.deterministic)
for fused in (False, True):
    agent = PPO()
    agent.actor.load_state_dict(det_weights(agent.actor, 1))
    agent.critic.load_state_dict(det_weights(agent.critic, 2))
    agent.to(DEV)
    if fused:
        agent.use_nhwc()
    s, p, a, gg, r, old = agent._unpack(buf, torch.device(DEV))
    adv, tv = agent.targets(s, p, gg, r)
    idx = torch.as_tensor(g["upd_perms"][0][:16]).to(DEV)
    agent.actor.train(); agent.critic.train()
    # one minibatch_step with lr 0 would still step Adam; read the gradients instead
    for o in (agent.optimizer_actor, agent.optimizer_critic):
        for grp in o.param_groups:
            grp["lr"] = 0.0
    agent.minibatch_step(s[idx][:, 0:4], p[idx][:, 0:4], gg[idx], a[idx], old[idx], adv[idx], tv[idx])
    grads[fused] = {n: prm.grad.detach().clone() for n, prm in list(agent.actor.named_parameters()) + [("c." + n, q) for n, q in agent.critic.named_parameters()]}
for n in grads[False]:
    a_, b_ = grads[False][n].double(), grads[True][n].double()
    if a_.shape != b_.shape:
        print(n, "shape differs", tuple(a_.shape), tuple(b_.shape)); continue
    den = float(a_.norm()) + 1e-30
    print("%-28s |g| %.3e  rel diff %.2e  max abs %.2e" % (n, den, float((a_ - b_).norm()) / den, float((a_ - b_).abs().max())))
