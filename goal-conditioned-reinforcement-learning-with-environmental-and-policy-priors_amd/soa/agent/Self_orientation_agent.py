"""Self-orientation agent (SoA) -- drop-in for the reference's soa/agent/Self_orientation_agent.py:38-294.

PPO + frozen world model (ppo_predictor) plus an orientation head: `agent_position_preditor` (sic, the
reference's attribute name) predicts where the agent will be three steps ahead as two 7-way distributions
over the offsets -3..3 of (y, x); the sampled offset is appended to the goal of actor and critic (4-d goal).
Two learners: `update_policy` (clipped PPO on actor / critic, :166-239) and `update_orientation` (negative
log-likelihood of the realised 3-step displacement, :242-294).

Sampling, TD targets and the PPO losses run on the HIP kernels (ppo_ops); the orientation NLL is two gathers
on torch.distributions.Categorical, as in the reference.
"""
import numpy as np
import torch
from torch.distributions import Categorical

from ... import ppo_ops
from .net.all_net import LSTM, Net_Decoder, Net_Encoder, Net_SoA_actor, Net_SoA_critic, Net_SoA_orient
from .PPO_Predictor import ppo_predictor


class self_orinetation_agent(ppo_predictor):
    def __init__(self, log_root=None, use_tensorboard=False):
        super().__init__(log_root=log_root, use_tensorboard=use_tensorboard)
        # construction order of the reference (:46-51)
        self.actor = Net_SoA_actor()
        self.critic = Net_SoA_critic()
        self.agent_position_preditor = Net_SoA_orient()
        self.encoder = Net_Encoder()
        self.decoder = Net_Decoder()
        self.predictor = LSTM()
        self.batch_size_pre_agent = 128
        self.K_epochs_pre_agent_position = 50
        self.update_count_fp = 0
        self.future3positionfilename = None
        self.grad_sync_orient = None               # multi-GPU: all-reduce of the orientation head's gradients (dist.py)
        self.optimizer_actor = torch.optim.Adam(self.actor.parameters(), lr=self.lr, eps=1e-5)
        self.optimizer_critic = torch.optim.Adam(self.critic.parameters(), lr=self.lr, eps=1e-5)
        self.optimizer_agent_position_preditor = torch.optim.Adam(self.agent_position_preditor.parameters(), lr=self.lr,
                                                                  eps=1e-5)
        mk = torch.optim.lr_scheduler.StepLR
        self.scheduler_actor = mk(self.optimizer_actor, self.lr_step_size, self.lr_gamma)
        self.scheduler_critic = mk(self.optimizer_critic, self.lr_step_size, self.lr_gamma)
        self.scheduler_agent_position_preditor = mk(self.optimizer_agent_position_preditor, 5 * self.lr_step_size,
                                                    self.lr_gamma)

    def to(self, device):
        for m in (self.actor, self.critic, self.agent_position_preditor, self.encoder, self.decoder, self.predictor):
            m.to(device)
        return self

    def trainable_nets(self):
        return [self.actor, self.critic, self.agent_position_preditor]

    # ------------------------------------------------------------------ acting
    def orient_probs(self, x8, p4, goal):
        """(Py_prob, Px_prob)-style pair of the reference: 7-way distributions of the two position components."""
        if self.amp_dtype is None:
            return self.agent_position_preditor(x8, p4, goal)
        with torch.autocast(device_type="cuda", dtype=self.amp_dtype):
            a, b = self.agent_position_preditor(x8, p4, goal)
        return a.float(), b.float()

    @torch.no_grad()
    def act_batch_soa(self, frames4, pos4, goal, uniforms=None, future=None, orient_only=False):
        """-> (action int32[B], logp f32[B], future f32[B,2]): orientation sample first, then the policy with the
        goal extended by it (:122-137).  uniforms f32[B,3] (component 0, component 1, action) or None = Philox.
        `future` given: that orientation sample is used instead of drawing one; orient_only: no action is drawn."""
        for m in (self.actor, self.critic, self.agent_position_preditor):
            m.eval()
        B = frames4.shape[0]
        x8 = self.policy_input(frames4)
        u = (None, None, None) if uniforms is None else tuple(uniforms[:, k].contiguous() for k in range(3))
        if future is None:
            p0, p1 = self.orient_probs(x8, pos4, goal)
            i0, _ = ppo_ops.sample(p0, u[0], seed=self.sample_seed, offset=self.sample_count)
            i1, _ = ppo_ops.sample(p1, u[1], seed=self.sample_seed, offset=self.sample_count + B)
            future = torch.stack([i0, i1], dim=1).float() - 3.0
        if orient_only:
            self.sample_count += 3 * B
            return None, None, future
        probs = self.actor_probs(x8, pos4, torch.cat([goal, future], dim=1))
        a, logp = ppo_ops.sample(probs, u[2], seed=self.sample_seed, offset=self.sample_count + 2 * B)
        self.sample_count += 3 * B
        return a, logp, future

    def act_batch(self, frames4, pos4, goal, uniforms=None):
        a, logp, self.last_future = self.act_batch_soa(frames4, pos4, goal, uniforms)
        return a, logp

    def select_action(self, state_matrix, states_stack, goal, device):
        """Reference signature (:105-144): -> (action, a_logp, future_x, future_y)."""
        sm = torch.as_tensor(np.asarray(state_matrix)[1:5], dtype=torch.float32, device=device).unsqueeze(0)
        st = torch.as_tensor(np.asarray(states_stack)[1:5], dtype=torch.float32, device=device).unsqueeze(0)
        g = torch.as_tensor(np.asarray(goal), dtype=torch.float32, device=device).unsqueeze(0)
        a, logp, f = self.act_batch_soa(sm, st, g)
        return int(a.item()), float(logp.item()), int(f[0, 0].item()), int(f[0, 1].item())

    # ------------------------------------------------------------------ learning
    @torch.no_grad()
    def _values(self, x8, p4, goal4, chunk=8192):
        v = torch.empty(x8.shape[0], device=x8.device)
        for i in range(0, x8.shape[0], chunk):
            j = min(x8.shape[0], i + chunk)
            v[i:j] = self.critic_value(x8[i:j], p4[i:j], goal4[i:j]).view(-1)
        return v

    def update_policy(self, buffer, device, i_ep, permutations=None):
        """Reference signature (:166-239); `buffer` = structured array of 9-frame window records
        (train_SoA.py:113-117): index 3 of the window is the acting state, index 4 the next one; a / r / a_logp / f
        column 0 belong to that step and f column 1 to the next."""
        device = torch.device(device)
        t = lambda k, dt=torch.float32: torch.as_tensor(buffer[k], dtype=dt, device=device)      # noqa: E731
        s, p, g, f = t("s"), t("p"), t("g"), t("f")
        a = t("a", torch.int64)[:, 0].view(-1).to(torch.int32)
        r = t("r")[:, 0].view(-1)
        old_logp = t("a_logp")[:, 0].view(-1, 1)
        n = s.shape[0]
        self.to(device)
        self.critic.eval()
        with torch.no_grad():
            x_next = self.policy_input(s[:, 1:5])
            x_cur = self.policy_input(s[:, 0:4])
            goal_next, goal_cur = torch.cat([g, f[:, 1]], 1), torch.cat([g, f[:, 0]], 1)
            nv = self._values(x_next, p[:, 1:5], goal_next)
            v = self._values(x_cur, p[:, 0:4], goal_cur)
            adv, target_v, _ = ppo_ops.gae(r.view(1, n).contiguous(), v.view(1, n), nv.view(1, n), None,
                                           gamma=self.gamma, lam=0.0, use_done_mask=False, want_ret=False)
            adv, target_v = adv.view(n, 1), target_v.view(n, 1)
        self.actor.train(); self.critic.train(); self.agent_position_preditor.train()
        la = lv = None
        for ep in range(self.K_epochs):
            perm = (torch.randperm(n) if permutations is None else torch.as_tensor(permutations[ep])).to(device)
            for i in range(0, n, self.batch_size):
                idx = perm[i:i + self.batch_size]
                la, lv = self.minibatch_step_x(x_cur[idx], p[idx][:, 0:4], goal_cur[idx], a[idx], old_logp[idx], adv[idx],
                                               target_v[idx])
        if la is not None:
            self.writer.add_scalar("loss/action_loss_i_ep", la, i_ep)
            self.writer.add_scalar("loss/value_loss_i_ep", lv, i_ep)
        if self.use_lr_decay:
            self.scheduler_actor.step()
            self.scheduler_critic.step()

    def orientation_step(self, x8, p4, goal, displacement, n_valid=None):
        """One optimiser step of the orientation head: NLL of the realised displacement (+3 -> class), :266-281.
        n_valid: only the first n_valid rows are samples, the rest pads the minibatch to a fixed shape (no loss)."""
        p0, p1 = self.orient_probs(x8, p4, goal)
        cls = (displacement + 3).long()
        nll = (-Categorical(probs=p0).log_prob(cls[:, 0]).view(-1, 1)
               - Categorical(probs=p1).log_prob(cls[:, 1]).view(-1, 1))
        loss = nll.mean() if n_valid is None else nll[:n_valid].sum() / float(n_valid)
        bucket = self.grad_sync_orient if hasattr(self.grad_sync_orient, "reduce_async") else None
        if bucket is not None:                       # dist.GradBucket: gradients accumulate straight into the bucket
            bucket.zero()
            loss.backward()
            bucket()
        else:
            self.optimizer_agent_position_preditor.zero_grad()
            loss.backward()
            if self.grad_sync_orient is not None:
                self.grad_sync_orient(list(self.agent_position_preditor.parameters()))
        if self.use_grad_clip:
            torch.nn.utils.clip_grad_norm_(self.agent_position_preditor.parameters(), 0.5)
        self.optimizer_agent_position_preditor.step()
        self.writer.add_scalar("loss/future_3steps_loss_update", loss.detach(), self.update_count_fp)
        self.update_count_fp += 1
        return loss.detach()

    def orientation_idle_step(self):
        """Optimiser step of a rank that has no orientation sample this update: zero local gradients, the same
        all-reduce and Adam step as its peers, so replicas stay identical and no collective is skipped."""
        if hasattr(self.grad_sync_orient, "reduce_async"):
            self.grad_sync_orient.zero()
            self.grad_sync_orient()
        else:
            self.optimizer_agent_position_preditor.zero_grad()
            for p in self.agent_position_preditor.parameters():
                p.grad = torch.zeros_like(p)
            if self.grad_sync_orient is not None:
                self.grad_sync_orient(list(self.agent_position_preditor.parameters()))
        if self.use_grad_clip:
            torch.nn.utils.clip_grad_norm_(self.agent_position_preditor.parameters(), 0.5)
        self.optimizer_agent_position_preditor.step()
        self.update_count_fp += 1

    def update_orientation(self, buffer, device, i_ep, permutations=None):
        """Reference signature (:242-294): the target is the position three states after the acting one minus the
        acting one, p[:, 6] - p[:, 3], which the window construction keeps within -3..3."""
        device = torch.device(device)
        t = lambda k: torch.as_tensor(buffer[k], dtype=torch.float32, device=device)             # noqa: E731
        s, p, g = t("s"), t("p"), t("g")
        n = s.shape[0]
        self.to(device)
        with torch.no_grad():
            x8 = self.policy_input(s[:, 0:4])
            disp = p[:, 6] - p[:, 3]
        self.agent_position_preditor.train()
        loss = None
        for ep in range(self.K_epochs_pre_agent_position):
            perm = (torch.randperm(n) if permutations is None else torch.as_tensor(permutations[ep])).to(device)
            for i in range(0, n, self.batch_size_pre_agent):
                idx = perm[i:i + self.batch_size_pre_agent]
                loss = self.orientation_step(x8[idx], p[idx][:, 0:4], g[idx], disp[idx])
        if loss is not None:
            self.writer.add_scalar("loss/future_3steps_loss_i_ep", loss, i_ep)
        if self.use_lr_decay:
            self.scheduler_agent_position_preditor.step()

    def save_param(self, i_ep, running_score):
        """Same keys as the reference (:146-153)."""
        import os
        from datetime import datetime
        state = {"model_actor": self.actor.state_dict(), "model_critic": self.critic.state_dict(),
                 "model_encoder": self.encoder.state_dict(), "model_decoder": self.decoder.state_dict(),
                 "model_predictor": self.predictor.state_dict(),
                 "model_agent_position_preditor": self.agent_position_preditor.state_dict(),
                 "optimizer_actor": self.optimizer_actor.state_dict(), "optimizer_critic": self.optimizer_critic.state_dict(),
                 "optimizer_agent_position_preditor": self.optimizer_agent_position_preditor.state_dict(), "epoch": i_ep}
        os.makedirs(self.filepath, exist_ok=True)
        path = os.path.join(str(self.filepath), "%s_net_%depoch_%srunning_score%s.pkl"
                            % (self.name, i_ep, running_score, datetime.now().strftime("%Y_%m_%d_%H_%M_%S")))
        torch.save(state, path)
        return path
