"""Vectorised PPO rollout / update loop on device-resident rollout tensors (replaces the per-transition
loop of the reference's soa/train_ppo.py:99-160 and the numpy ring buffer for the N-env path).

Rollout storage is time-major and holds ONE 289-float frame per env-step (not the reference's 5-frame
stack per record, train_ppo.py:93-97): frames[k] for k = -3..T, pos[k], age[t] (steps since the env's
reset).  The 4-frame policy input of any (t, n) is assembled on the fly by ppo_gather_stack, which
repeats the reset frame at episode starts exactly like np.tile in Env_transact.reset.
"""
import torch

from .. import ppo_ops

INIT_POS = (15.0, 3.0)        # agent (y, x) after reset (twoarmy_v6.py:10, env_buffer.py:320-322)
GOAL_YX = (2.0, 14.0)


class VecPPOTrainer:
    def __init__(self, agent, engine, rollout_steps=128, minibatch=4096, value_chunk=16384, frame_codes=False):
        """frame_codes=True stores the rollout's frames as uint8 codes (TW_F_MATRIX_CODE; 304 B instead of 1168 B
        per env-step) and expands them to the same fp32 policy inputs when stacks are gathered: bit-identical
        training, 3.8x less frame memory (BASELINE config 5)."""
        self.agent, self.engine = agent, engine
        self.T, self.N = int(rollout_steps), engine.num_envs
        self.minibatch, self.value_chunk = int(minibatch), int(value_chunk)
        d = self.device = engine.device
        T, N = self.T, self.N
        self.frame_codes = bool(frame_codes)
        if self.frame_codes:
            self.frames_buf = torch.zeros((T + 4, N, 304), dtype=torch.uint8, device=d)
        else:
            self.frames_buf = torch.zeros((T + 4, N, 292), dtype=torch.float32, device=d)
        self.frames = self.frames_buf[..., :289]
        self.pos = torch.zeros((T + 4, N, 2), dtype=torch.float32, device=d)
        self.action = torch.zeros((T, N), dtype=torch.int32, device=d)
        self.logp = torch.zeros((T, N), dtype=torch.float32, device=d)
        self.reward = torch.zeros((T, N), dtype=torch.float32, device=d)
        self.term = torch.zeros((T, N), dtype=torch.uint8, device=d)
        self.trunc = torch.zeros((T, N), dtype=torch.uint8, device=d)
        self.age = torch.zeros((T + 1, N), dtype=torch.int32, device=d)
        self.goal1 = torch.tensor([GOAL_YX], device=d)
        self.goal = self.goal1.expand(N, 2).contiguous()
        self.n_all = torch.arange(N, dtype=torch.int32, device=d)
        self.init_pos = torch.tensor(INIT_POS, device=d)
        # the reset frame is a constant of the task: take it from a scratch engine step-free reset
        self.init_frame = self._reset_frame()
        self.frames[:4] = self._encode(self.init_frame) if self.frame_codes else self.init_frame
        self.pos[:4] = self.init_pos
        agent.to(d)
        self.env_steps = 0
        self.episodes_done = 0
        self.return_sum = 0.0

    @staticmethod
    def _encode(frame):
        """float matrix -> codes (0 free/goal 0.9, 1 wall -0.9, 2 ball -0.5, 3 agent 0.3)."""
        c = torch.zeros_like(frame, dtype=torch.uint8)
        c[frame == -0.9] = 1
        c[frame == -0.5] = 2
        c[frame == 0.3] = 3
        return c

    def _reset_frame(self):
        ty, _, rec = self.engine.get_state()
        m = torch.where(torch.tensor(ty[0] == 2), -0.9, torch.where(torch.tensor(ty[0] == 6), -0.5, 0.9)).float()
        m[15 * 17 + 3] = 0.3
        # valid only for freshly reset envs; the engine was just created / reset by the caller
        return m.to(self.device)

    # ------------------------------------------------------------------ rollout
    @torch.no_grad()
    def collect(self, uniforms=None):
        T, N = self.T, self.N
        for t in range(T):
            k = torch.full((N,), t + 3, dtype=torch.int32, device=self.device)
            s4, p4 = ppo_ops.gather_stack(self.frames, self.pos, k, self.n_all, self.age[t], self.init_frame,
                                          self.init_pos)
            a, logp = self.agent.act_batch(s4, p4, self.goal, None if uniforms is None else uniforms[t])
            self.action[t], self.logp[t] = a, logp
            out = {"obs": None, "matrix": self.frames[t + 4], "pos": self.pos[t + 4], "reward": self.reward[t],
                   "terminated": self.term[t], "truncated": self.trunc[t]}
            self.engine.step(a, out, autoreset=True, policy_idx=True)
            done = (self.term[t] | self.trunc[t]) != 0
            self.age[t + 1] = torch.where(done, torch.zeros_like(self.age[t]), self.age[t] + 1)
        self.env_steps += T * N

    def carry_over(self):
        """Make the last 4 frames the history of the next rollout."""
        T = self.T
        self.frames_buf[:4] = self.frames_buf[T:T + 4].clone()
        self.pos[:4] = self.pos[T:T + 4].clone()
        self.age[0] = self.age[T]

    # ------------------------------------------------------------------ update
    def _stacks(self, t_idx, n_idx, after):
        """4-frame stacks of samples (t, n): before the step (s[:,0:4]) or after it (s[:,1:5])."""
        k = t_idx + (4 if after else 3)
        age = self.age[:-1].reshape(-1)[t_idx.long() * self.N + n_idx.long()] + (1 if after else 0)
        return ppo_ops.gather_stack(self.frames, self.pos, k.int(), n_idx.int(), age.int(), self.init_frame,
                                    self.init_pos)

    @torch.no_grad()
    def compute_targets(self):
        T, N = self.T, self.N
        total = T * N
        v = torch.empty(total, device=self.device)
        nv = torch.empty(total, device=self.device)
        idx = torch.arange(total, device=self.device)
        self.agent.critic.eval()
        for i in range(0, total, self.value_chunk):
            sl = idx[i:i + self.value_chunk]
            t_idx, n_idx = sl // N, sl % N
            g = self.goal1.expand(sl.numel(), 2)
            s0, p0 = self._stacks(t_idx, n_idx, after=False)
            v[i:i + sl.numel()] = self.agent.critic(self.agent.policy_input(s0), p0, g).view(-1)
            s1, p1 = self._stacks(t_idx, n_idx, after=True)
            nv[i:i + sl.numel()] = self.agent.critic(self.agent.policy_input(s1), p1, g).view(-1)
        done = (self.term | self.trunc).contiguous()
        adv, target, ret = ppo_ops.gae(self.reward, v.view(T, N), nv.view(T, N), done, gamma=self.agent.gamma,
                                       lam=self.agent.gae_lambda, use_done_mask=self.agent.use_done_mask)
        critic_target = target if self.agent.gae_lambda == 0.0 else ret
        if self.agent.normalize_adv:
            ppo_ops.adv_norm_(adv)
        return adv, critic_target

    def update(self, permutations=None):
        ag = self.agent
        T, N = self.T, self.N
        total = T * N
        adv, target = self.compute_targets()
        adv, target = adv.view(-1), target.view(-1)
        act, logp = self.action.view(-1), self.logp.view(-1)
        ag.actor.train(); ag.critic.train()
        la = lv = None
        for ep in range(ag.K_epochs):
            perm = (torch.randperm(total) if permutations is None else torch.as_tensor(permutations[ep])).to(self.device)
            for i in range(0, total, self.minibatch):
                idx = perm[i:i + self.minibatch]
                t_idx, n_idx = idx // N, idx % N
                s0, p0 = self._stacks(t_idx, n_idx, after=False)
                la, lv = ag.minibatch_step(s0, p0, self.goal1.expand(idx.numel(), 2), act[idx], logp[idx].view(-1, 1),
                                           adv[idx].view(-1, 1), target[idx].view(-1, 1))
        if ag.use_lr_decay:
            ag.scheduler_actor.step(); ag.scheduler_critic.step()
        return la, lv

    def stats(self):
        done = (self.term | self.trunc) != 0
        return {"env_steps": self.env_steps, "episodes": int(done.sum()), "successes": int(self.term.sum()),
                "mean_reward": float(self.reward.mean())}
