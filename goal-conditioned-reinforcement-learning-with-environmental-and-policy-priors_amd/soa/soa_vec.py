"""Vectorised trainer of the self-orientation agent (reference soa/train_SoA.py:123-272 +
Self_orientation_agent.py:166-294) on the time-major rollout of VecPPOTrainer.

The reference keeps 9-frame window records (3 history frames, the acting state at index 3, 5 future frames) so that
`update_orientation` can read the position three states ahead (p[:,6] - p[:,3]) and `update_policy` the
orientation sample of the next step (f[:,1]).  Here both are index arithmetic on the rollout: f is stored per step,
the next step's f is f[t+1] (or f[t] again when the episode ends at t, which is what the reference's four terminal
window shifts produce), and the 3-step displacement is pos[min(t+3, end+1)] - pos[t] inside the episode.
The orientation head trains, like the reference (train_SoA.py:206-224, 243-262), only on trajectories that reach
their goal: episodes that terminated, plus the hindsight records of the others.

Success replay (train_SoA.py:200-205, 243-262): the reference keeps the window records of the last 99 episodes that
reached the goal in a FIFO that survives its buffer resets, and trains the orientation head on that FIFO at every
update.  Here the FIFO holds materialised orientation samples (4-frame stacks, positions, goal, displacement) of
goal-reaching episodes of EARLIER rollouts; an update trains on every goal-reaching episode of the current rollout
plus the newest (99 - that number, if positive) episodes of the FIFO -- exactly the reference's "last 99 successes"
whenever the current rollout holds at most 99 of them (its scale), and never fewer samples than the rollout offers."""
import collections

import torch

from .. import ppo_ops
from .ppo_vec import VecPPOTrainer


class VecSoATrainer(VecPPOTrainer):
    def __init__(self, agent, engine, rollout_steps=128, minibatch=4096, value_chunk=16384, frame_codes=False,
                 orient_minibatch=None):
        super().__init__(agent, engine, rollout_steps, minibatch, value_chunk, frame_codes)
        T, N = self.T, self.N
        self.future = torch.zeros((T + 1, N, 2), dtype=torch.float32, device=self.device)     # f of the acting state t
        self.pending_future = None                       # f already drawn for the first state of the next rollout
        self.orient_minibatch = int(orient_minibatch or minibatch)
        self.replay_episodes = 99                        # len(fp_terminate_buffer) > 99 -> pop(0), train_SoA.py:204-205
        self._replay = collections.deque()               # oldest first; one dict of sample tensors per episode

    # ------------------------------------------------------------------ rollout
    @torch.no_grad()
    def collect(self, uniforms=None):
        """uniforms: f32[T+1, N, 3] or None (row T feeds the orientation draw of the state after the last step)."""
        T, N = self.T, self.N
        ag = self.agent
        self._pred_valid = False                         # this rollout loop does not fill the prediction cache
        for t in range(T + 1):
            k = torch.full((N,), t + 3, dtype=torch.int32, device=self.device)
            s4, p4 = ppo_ops.gather_stack(self.frames, self.pos, k, self.n_all, self.age[t], self.init_frame, self.init_pos)
            u = None if uniforms is None else uniforms[t]
            if t == T:                                   # only the orientation draw: it belongs to the next rollout's step 0
                _, _, self.future[T] = ag.act_batch_soa(s4, p4, self.goal, u, orient_only=True)
                break
            forced = self.pending_future if t == 0 else None
            a, logp, f = ag.act_batch_soa(s4, p4, self.goal, u, future=forced)
            self.action[t], self.logp[t], self.future[t] = a, logp, f
            out = {"obs": None, "matrix": self.frames[t + 4], "pos": self.pos[t + 4], "reward": self.reward[t],
                   "terminated": self.term[t], "truncated": self.trunc[t]}
            self.engine.step(a, out, autoreset=True, policy_idx=True)
            done = (self.term[t] | self.trunc[t]) != 0
            self.age[t + 1] = torch.where(done, torch.zeros_like(self.age[t]), self.age[t] + 1)
        self.pending_future = self.future[T].clone()
        self.env_steps += T * N

    # ------------------------------------------------------------------ goals seen by the networks
    def sample_goal(self, t_idx, n_idx, goal2, done):
        """[goal(2), f of the acting state(2), f of the next state(2)]; the next state's f repeats the current one
        when the (possibly relabelled) episode ends at this step."""
        t, n = t_idx.long(), n_idx.long()
        f_cur = self.future[t, n]
        f_next = torch.where((done != 0).view(-1, 1), f_cur, self.future[t + 1, n])
        return torch.cat([goal2, f_cur, f_next], dim=1)

    def goal_input(self, goal, after):
        return torch.cat([goal[:, 0:2], goal[:, 4:6] if after else goal[:, 2:4]], dim=1)

    # ------------------------------------------------------------------ orientation head
    @torch.no_grad()
    def orientation_samples(self):
        """(t, n, goal2, displacement) of the samples the orientation head trains on."""
        T, N = self.T, self.N
        dev = self.device
        done = ((self.term | self.trunc) != 0)
        tt = torch.arange(T, device=dev).view(T, 1).expand(T, N)
        big = torch.full((T, N), T + 8, device=dev, dtype=torch.long)
        end = torch.flip(torch.cummin(torch.flip(torch.where(done, tt, big), [0]), 0).values, [0])   # end step of the episode
        ended = end < T
        success = ended & (self.term.long().gather(0, end.clamp(max=T - 1)) != 0)
        t_idx, n_idx = torch.nonzero(success, as_tuple=True)
        u = end[t_idx, n_idx]
        goal2 = self.goal1.expand(t_idx.numel(), 2)
        self._n_success_samples = int(t_idx.numel())
        self._success_key = u * N + n_idx                                            # episode id: (end step, env)
        if self.her is not None and self.her["t"].numel():
            h = self.her
            hs = torch.nonzero(h["done"]).view(-1)                                   # last record of every relabelled prefix
            j = torch.searchsorted(hs, torch.arange(h["t"].numel(), device=dev))
            hu = h["t"].long()[hs[j]]
            # hindsight records of episodes that did NOT reach the real goal (train_SoA.py:217-219)
            keep = ~success[h["t"].long(), h["n"].long()]
            t_idx = torch.cat([t_idx, h["t"].long()[keep]])
            n_idx = torch.cat([n_idx, h["n"].long()[keep]])
            u = torch.cat([u, hu[keep]])
            goal2 = torch.cat([goal2, h["goal"][keep]])
        # state index s (before step s) of env n sits in self.pos[s + 3]; the first state of an episode is the reset state
        start = self.age[:-1].long()[t_idx, n_idx] == 0
        p_now = torch.where(start.view(-1, 1), self.init_pos.view(1, 2), self.pos[t_idx + 3, n_idx])
        p_fut = self.pos[torch.minimum(t_idx + 3, u + 1) + 3, n_idx]
        return t_idx.int(), n_idx.int(), goal2, p_fut - p_now

    # ------------------------------------------------------------------ success replay (train_SoA.py:200-205)
    @torch.no_grad()
    def replay_samples(self, n_current_episodes):
        """Materialised samples of the newest (replay_episodes - n_current_episodes) remembered episodes, or None."""
        k = max(0, self.replay_episodes - int(n_current_episodes))
        eps = list(self._replay)[-k:] if k else []
        if not eps:
            return None
        return {key: torch.cat([e[key] for e in eps]) for key in ("s0", "p0", "goal2", "disp")}

    @torch.no_grad()
    def remember_successes(self, t_idx, n_idx, goal2, disp):
        """Push the goal-reaching episodes of this rollout (the first _n_success_samples samples) into the FIFO in
        completion order (end step, then env); only the newest `replay_episodes` are kept."""
        ns = self._n_success_samples
        if ns == 0:
            return
        keys, inv = torch.unique(self._success_key, return_inverse=True)              # ascending = completion order
        first = max(0, keys.numel() - self.replay_episodes)
        sel = torch.nonzero(inv >= first).view(-1)
        s0, p0 = self._stacks(t_idx[:ns][sel], n_idx[:ns][sel], after=False)
        ep = (inv[sel] - first).cpu()
        order = torch.argsort(ep, stable=True)
        counts = torch.bincount(ep, minlength=keys.numel() - first).tolist()
        sel_o = order.to(self.device)
        s0, p0, g2, dp = s0[sel_o], p0[sel_o], goal2[:ns][sel][sel_o], disp[:ns][sel][sel_o]
        off = 0
        for c in counts:
            self._replay.append(dict(s0=s0[off:off + c].clone(), p0=p0[off:off + c].clone(),
                                     goal2=g2[off:off + c].clone(), disp=dp[off:off + c].clone()))
            off += c
        while len(self._replay) > self.replay_episodes:
            self._replay.popleft()

    def update_orientation(self, permutations=None):
        ag = self.agent
        t_idx, n_idx, goal2, disp = self.orientation_samples()
        n_rollout = t_idx.numel()
        n_cur_eps = int(torch.unique(self._success_key).numel()) if self._n_success_samples else 0
        rep = self.replay_samples(n_cur_eps)
        n_rep = 0 if rep is None else int(rep["disp"].shape[0])
        total = n_rollout + n_rep                            # sample ids: [0, n_rollout) rollout, then the replay
        local_steps = n_steps = -(-total // self.orient_minibatch)
        synced = ag.grad_sync_orient is not None and torch.distributed.is_initialized()
        if synced:
            # EVERY rank joins this collective, also one without a single orientation sample (no success, no
            # hindsight record): it then takes part in each gradient all-reduce with zero gradients
            from .ppo_vec import agree_on_steps
            n_steps = agree_on_steps(n_steps, self.device)
        if n_steps == 0:
            return None
        ag.agent_position_preditor.train()
        loss = None
        for ep in range(ag.K_epochs_pre_agent_position):
            if total == 0:
                for _ in range(n_steps):
                    ag.orientation_idle_step()
                continue
            perm = (torch.randperm(total) if permutations is None else torch.as_tensor(permutations[ep])).to(self.device)
            if synced and n_steps > -(-perm.numel() // self.orient_minibatch):
                perm = perm[torch.arange(n_steps * self.orient_minibatch, device=self.device) % perm.numel()]
            done_steps = 0
            for i in range(0, perm.numel(), self.orient_minibatch):
                idx = perm[i:i + self.orient_minibatch]
                cur = idx[idx < n_rollout]
                s0, p0 = self._stacks(t_idx[cur], n_idx[cur], after=False)
                g2, dp = goal2[cur], disp[cur]
                if n_rep:
                    old = idx[idx >= n_rollout] - n_rollout
                    s0, p0 = torch.cat([s0, rep["s0"][old]]), torch.cat([p0, rep["p0"][old]])
                    g2, dp = torch.cat([g2, rep["goal2"][old]]), torch.cat([dp, rep["disp"][old]])
                n_valid = None
                if self.fixed_shapes and idx.numel() < self.orient_minibatch:
                    # partial minibatch: padded to the full shape (or, for a sample set smaller than one minibatch, to
                    # the next power of two) with masked rows behind the real ones -- the sample count changes every
                    # update and every new conv batch size costs a MIOpen kernel search
                    full = self.orient_minibatch if perm.numel() >= self.orient_minibatch else \
                        max(64, 1 << (idx.numel() - 1).bit_length())
                    if full > idx.numel():
                        n_valid = idx.numel()
                        pad = torch.arange(full - n_valid, device=self.device) % n_valid
                        s0, p0 = torch.cat([s0, s0[pad]]), torch.cat([p0, p0[pad]])
                        g2, dp = torch.cat([g2, g2[pad]]), torch.cat([dp, dp[pad]])
                with torch.no_grad():
                    x8 = ag.policy_input(s0)
                loss = ag.orientation_step(x8, p0, g2, dp, n_valid)
                done_steps += 1
            assert done_steps == n_steps or not synced, (done_steps, n_steps)
        if ag.use_lr_decay:
            ag.scheduler_agent_position_preditor.step()
        self.remember_successes(t_idx, n_idx, goal2, disp)
        return loss

    def update(self, permutations=None, orient_permutations=None):
        """update_policy, then update_orientation (train_SoA.py:264-265); the hindsight records serve both."""
        her = self.her
        la, lv = super().update(permutations)
        self.her = her
        lo = self.update_orientation(orient_permutations)
        self.her = None
        self.last_orientation_loss = lo
        return la, lv
