"""PPO learner -- drop-in for the reference's soa/agent/PPO.py on top of the HIP PPO kernels.

Same public surface (reference PPO.py:41-161): attributes actor / critic / optimizer_* / gamma /
clip_param / K_epochs / entropy_coef / batch_size / use_grad_clip / use_lr_decay / writer,
`select_action(state_matrix, states_stack, goal, device) -> (int, float)`,
`update(buffer, device, i_ep)`, `save_param(i_ep, running_score)`.

What runs where:
  * actor / critic conv + linear GEMMs: PyTorch-ROCm (MFMA through MIOpen / hipBLASLt);
  * Categorical sample / log-prob, TD target + advantage, the clipped-surrogate + entropy loss and the
    SmoothL1 critic loss with their backward: hand-written HIP (ppo_ops, csrc/ppo_kernels.hip);
  * Adam(lr 1e-4, eps 1e-5) x 2 and StepLR(200, 0.8): torch.optim, as in the reference.
There is no CPU fallback for the kernels: `device` must be a GPU.
"""
import os
from datetime import datetime
from pathlib import Path

import numpy as np
import torch

from ... import ppo_ops
from .net.all_net import Net_PPO_actor, Net_PPO_critic


class ScalarLog:
    """Minimal stand-in for tensorboardX.SummaryWriter (the reference's parity probes are its scalars).
    Values may be device tensors; they are only synchronised when read through `scalars`."""

    def __init__(self, log_dir=None):
        self.log_dir = log_dir
        self._raw = {}

    def add_scalar(self, tag, value, step=None):
        self._raw.setdefault(tag, []).append((step, value))

    @property
    def scalars(self):
        return {k: [(s, float(v)) for s, v in vs] for k, vs in self._raw.items()}

    def close(self):
        pass


def _make_writer(log_dir):
    try:                                   # use the real thing when it is installed
        from tensorboardX import SummaryWriter
        return SummaryWriter(log_dir=log_dir)
    except Exception:
        return ScalarLog(log_dir)


class PPO:
    def __init__(self, log_root=None, use_tensorboard=False):
        self.traindate = datetime.now().strftime("%Y_%m_%d_%H_%M_%S")
        root = Path(log_root or os.environ.get("TWOARMY_LOG_ROOT", "./runs"))
        self.filepath = root / "param" / "PPO" / ("gridim_ppo" + self.traindate)
        self.writer_dir = str(root / "tensor" / "PPO" / ("logs_" + self.traindate))
        self.writer = _make_writer(self.writer_dir) if use_tensorboard else ScalarLog(self.writer_dir)
        self.actor = Net_PPO_actor()               # same order as the reference: actor, then critic
        self.critic = Net_PPO_critic()
        self.gamma = 0.99
        self.lr = 0.0001
        self.weight_decay = 0.0001                 # unused by the reference as well (PPO.py:53)
        self.lr_step_size = 200
        self.lr_gamma = 0.8
        self.batch_size = 128
        self.optimizer_actor = torch.optim.Adam(self.actor.parameters(), lr=self.lr, eps=1e-5)
        self.optimizer_critic = torch.optim.Adam(self.critic.parameters(), lr=self.lr, eps=1e-5)
        self.scheduler_actor = torch.optim.lr_scheduler.StepLR(self.optimizer_actor, self.lr_step_size, self.lr_gamma)
        self.scheduler_critic = torch.optim.lr_scheduler.StepLR(self.optimizer_critic, self.lr_step_size, self.lr_gamma)
        self.update_count = 0
        self.clip_param = 0.1
        self.K_epochs = 10
        self.entropy_coef = 0.01
        self.use_grad_clip = False
        self.use_lr_decay = False
        self.max_steps = 0
        self.heatmapfilename = None
        self.name = None
        self.sample_seed = 9981                    # Philox key of the on-device action sampler
        self.sample_count = 0
        # north-star extensions (no reference counterpart; the defaults reproduce the reference)
        self.gae_lambda = 0.0
        self.use_done_mask = False
        self.normalize_adv = False
        self.grad_sync = None                      # multi-GPU: dist.GradBucket([actor params, critic params]) (or any callable(params))
        self.amp_dtype = None                      # torch.bfloat16: conv/linear GEMMs on bf16 MFMA (opt-in; fp32 = parity)

    # ------------------------------------------------------------------ acting
    def policy_input(self, frames4):
        """Hook: what the networks consume for a 4-frame stack (the predictor variant appends 4 predicted frames)."""
        return frames4

    def _run(self, net, x, p, g):
        """Network forward; under `amp_dtype` the conv/linear GEMMs run in that dtype (fp32 master weights, fp32
        outputs and losses).  Default (None) is plain fp32, the only mode with a parity claim."""
        if self.amp_dtype is None:
            return net(x, p, g)
        with torch.autocast(device_type="cuda", dtype=self.amp_dtype):
            out = net(x, p, g)
        return out.float()

    def actor_probs(self, x, p, g):
        return self._run(self.actor, x, p, g)

    def critic_value(self, x, p, g):
        return self._run(self.critic, x, p, g)

    @torch.no_grad()
    def act_batch(self, frames4, pos4, goal, uniforms=None, offset_dev=None, offset_add=0, x=None):
        """frames4 [B,4,289], pos4 [B,4,2], goal [B,2] (device) -> (action int32[B], logp float[B]).
        offset_dev (int64[1] device tensor): the sampler's stream position is *offset_dev + offset_add instead of
        self.sample_count -- for launches recorded in a HIP graph, whose position must live in device memory.
        x: policy_input(frames4) when the caller has already computed it."""
        self.actor.eval()
        probs = self.actor_probs(self.policy_input(frames4) if x is None else x, pos4, goal)
        if offset_dev is not None:
            return ppo_ops.sample(probs, uniforms, seed=self.sample_seed, offset=int(offset_add), offset_dev=offset_dev)
        a, logp = ppo_ops.sample(probs, uniforms, seed=self.sample_seed, offset=self.sample_count)
        self.sample_count += probs.shape[0]
        return a, logp

    def to(self, device):
        self.actor.to(device); self.critic.to(device)
        return self

    def trainable_nets(self):
        return [self.actor, self.critic]

    def use_nhwc(self, enable=True):
        """Channels-last conv stacks + fused conv epilogues (all_net.TINet.nhwc): the fast layout on MI355X (no MIOpen
        transposes, MFMA backward-data kernels).  Same arithmetic up to the summation order inside the conv GEMMs; the
        default (NCHW, plain nn.Sequential) stays the literal parity mode.  Call after .to(device)."""
        from .net.all_net import use_nhwc
        use_nhwc(self.trainable_nets(), enable)
        return self

    def select_action(self, state_matrix, states_stack, goal, device):
        """Reference signature (PPO.py:73-92): 5-deep numpy stacks in, python (action, log-prob) out."""
        sm = torch.as_tensor(np.asarray(state_matrix)[1:5], dtype=torch.float32, device=device).unsqueeze(0)
        st = torch.as_tensor(np.asarray(states_stack)[1:5], dtype=torch.float32, device=device).unsqueeze(0)
        g = torch.as_tensor(np.asarray(goal), dtype=torch.float32, device=device).unsqueeze(0)
        self.critic.eval()
        a, logp = self.act_batch(sm, st, g)
        return int(a.item()), float(logp.item())

    # ------------------------------------------------------------------ learning
    @torch.no_grad()
    def targets(self, s, p, g, r, done=None, chunk=8192):
        """target_v = r + gamma * V(s[:,1:5]); adv = target_v - V(s[:,0:4])  (PPO.py:112-115) via ppo_gae."""
        n = s.shape[0]
        v = torch.empty(n, device=s.device)
        nv = torch.empty(n, device=s.device)
        for i in range(0, n, chunk):
            j = min(n, i + chunk)
            nv[i:j] = self.critic_value(self.policy_input(s[i:j, 1:5]), p[i:j, 1:5], g[i:j]).view(-1)
            v[i:j] = self.critic_value(self.policy_input(s[i:j, 0:4]), p[i:j, 0:4], g[i:j]).view(-1)
        adv, target, _ = ppo_ops.gae(r.view(1, n).contiguous(), v.view(1, n), nv.view(1, n),
                                     None if done is None else done.view(1, n).contiguous(),
                                     gamma=self.gamma, lam=0.0, use_done_mask=False, want_ret=False)
        return adv.view(n, 1), target.view(n, 1)

    def minibatch_step(self, s0, p0, g, a, old_logp, adv, target_v, n_valid=None):
        """One optimiser step on one minibatch (PPO.py:122-147); returns (action_loss, value_loss) tensors.
        n_valid: the first n_valid rows are real samples, the rest pads the minibatch to a fixed shape."""
        return self.minibatch_step_x(self.policy_input(s0), p0, g, a, old_logp, adv, target_v, n_valid)

    def minibatch_step_x(self, x0, p0, g, a, old_logp, adv, target_v, n_valid=None):
        """minibatch_step on already assembled network inputs (frames incl. any predicted ones)."""
        probs = self.actor_probs(x0, p0, g)
        value = self.critic_value(x0, p0, g)
        action_loss, value_loss = ppo_ops.ppo_losses(probs, value, a, old_logp, adv, target_v,
                                                     clip=self.clip_param, ent_coef=self.entropy_coef, n_valid=n_valid)
        bucket = self.grad_sync if hasattr(self.grad_sync, "reduce_async") else None
        if bucket is not None:
            # multi-GPU, dist.GradBucket([actor params, critic params]): gradients are views into one flat buffer; the
            # actor's all-reduce is in flight while the critic's backward computes
            split = len(bucket.parts) == 2               # [actor, critic]; a one-group bucket is reduced after both
            bucket.zero()
            action_loss.backward()
            if split:
                bucket.reduce_async(0)
            value_loss.backward()
            bucket.reduce_async(1 if split else None)
            bucket.finish()
        else:
            self.optimizer_actor.zero_grad()
            self.optimizer_critic.zero_grad()
            action_loss.backward()
            value_loss.backward()
            if self.grad_sync is not None:
                self.grad_sync(list(self.actor.parameters()) + list(self.critic.parameters()))
        if self.use_grad_clip:
            torch.nn.utils.clip_grad_norm_(self.actor.parameters(), 0.5)
            torch.nn.utils.clip_grad_norm_(self.critic.parameters(), 0.5)
        self.optimizer_actor.step()
        self.optimizer_critic.step()
        self.writer.add_scalar("loss/action_loss_update", action_loss.detach(), self.update_count)
        self.writer.add_scalar("loss/value_loss_update", value_loss.detach(), self.update_count)
        self.update_count += 1
        return action_loss.detach(), value_loss.detach()

    def _unpack(self, buffer, device):
        """Record array of Buffer_gridworld -> tensors of the transition every record trains on: 5-frame stacks
        s[n,5,289] / p[n,5,2] (acting state = frames 0..3, next state = frames 1..4), a[n], g[n,2], r[n], old_logp[n,1]."""
        f32 = lambda k: torch.as_tensor(buffer[k], dtype=torch.float32, device=device)
        a = torch.as_tensor(buffer["a"], dtype=torch.int64, device=device).view(-1).to(torch.int32)
        return f32("s"), f32("p"), a, f32("g"), f32("r").view(-1), f32("a_logp").view(-1, 1)

    def update(self, buffer, device, i_ep, permutations=None):
        """Reference signature (PPO.py:103-161): `buffer` is the numpy structured array of
        Buffer_gridworld.  `permutations` (optional, [K_epochs][N]) injects the minibatch order;
        by default torch.randperm is drawn per epoch exactly like SubsetRandomSampler does."""
        device = torch.device(device)
        s, p, a, g, r, old_logp = self._unpack(buffer, device)
        n = s.shape[0]
        self.to(device)
        adv, target_v = self.targets(s, p, g, r)
        if self.normalize_adv:
            ppo_ops.adv_norm_(adv)
        self.actor.train(); self.critic.train()
        la = lv = None
        for ep in range(self.K_epochs):
            perm = torch.randperm(n) if permutations is None else torch.as_tensor(permutations[ep])
            perm = perm.to(device)
            for i in range(0, n, self.batch_size):
                idx = perm[i:i + self.batch_size]
                la, lv = self.minibatch_step(s[idx][:, 0:4], p[idx][:, 0:4], g[idx], a[idx], old_logp[idx],
                                             adv[idx], target_v[idx])
        if la is not None:
            self.writer.add_scalar("loss/action_loss_i_ep", la, i_ep)
            self.writer.add_scalar("loss/value_loss_i_ep", lv, i_ep)
        if self.use_lr_decay:
            self.scheduler_actor.step()
            self.scheduler_critic.step()

    # ------------------------------------------------------------------ checkpoints
    def save_param(self, i_ep, running_score):
        """Same dict layout / key names as the reference (PPO.py:94-100)."""
        state = {"model_actor": self.actor.state_dict(), "model_critic": self.critic.state_dict(),
                 "optimizer_actor": self.optimizer_actor.state_dict(),
                 "optimizer_critic": self.optimizer_critic.state_dict(), "epoch": i_ep}
        os.makedirs(self.filepath, exist_ok=True)
        path = os.path.join(str(self.filepath), "%s_net_%depoch_%srunning_score%s.pkl"
                            % (self.name, i_ep, running_score, datetime.now().strftime("%Y_%m_%d_%H_%M_%S")))
        torch.save(state, path)
        return path
