"""Vectorised PPO rollout / update loop on device-resident rollout tensors (replaces the per-transition
loop of the reference's soa/train_ppo.py:99-160 and the numpy ring buffer for the N-env path).

Rollout storage is time-major and holds ONE 289-float frame per env-step (not the reference's 5-frame
stack per record, train_ppo.py:93-97): frames[k] for k = -3..T, pos[k], age[t] (steps since the env's
reset).  The 4-frame policy input of any (t, n) is assembled on the fly by ppo_gather_stack, which
repeats the reset frame at episode starts exactly like np.tile in Env_transact.reset.
"""
import collections
import time

import torch

from .. import ppo_ops

HER_MAX_LEN = 64              # longest episode ppo_her_relabel handles (csrc/ppo_kernels.hip)

INIT_POS = (15.0, 3.0)        # agent (y, x) after reset (twoarmy_v6.py:10, env_buffer.py:320-322)
GOAL_YX = (2.0, 14.0)


def agree_on_steps(local_steps, device):
    """Optimiser steps per epoch every rank will take: the MAX over ranks of ceil(samples / minibatch) (relabelled-record
    counts differ per rank, and every rank must take part in the same number of gradient all-reduces)."""
    if not (torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1):
        return int(local_steps)
    on_cpu = torch.distributed.get_backend() != "nccl"
    m = torch.tensor([int(local_steps)], device="cpu" if on_cpu else device)
    torch.distributed.all_reduce(m, op=torch.distributed.ReduceOp.MAX)
    return int(m.item())


def padded_permutation(perm, n_steps, minibatch):
    """An epoch's index stream stretched to exactly n_steps minibatches: a rank with fewer samples than its peers
    revisits samples of this epoch (cyclically) so that it issues as many gradient all-reduces as every other rank."""
    if n_steps > -(-perm.numel() // minibatch):
        perm = perm[torch.arange(n_steps * minibatch, device=perm.device) % perm.numel()]
    return perm


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
        self.reuse_next_values = True             # V(s'_t) = V(s_{t+1}) inside an episode (_values_rollout)
        self.fixed_shapes = True                  # pad partial minibatches / value chunks to the full size (masked rows)
        # agents whose policy input holds frames PREDICTED by a frozen world model (ppo_predictor): those frames depend only
        # on the acting state's 4-frame stack, so the rollout's own evaluation is kept ([T][N][4][289], 2.4 GB at
        # 4096 x 128) and the target pass and every epoch reuse it instead of re-running the encoder -> LSTM -> decoder
        # (7x the flops of the policy itself) on the same states
        self.cache_predictions = hasattr(agent, "pred_states")
        self.pred_frames, self._pred_valid = None, False
        self.use_graph = False                    # collect(): replay the rollout as one HIP graph (see collect)
        self._graph, self._graph_warm = None, False
        self._graph_base = torch.zeros(1, dtype=torch.int64, device=d)
        self.time_phases = False                  # update(): wall time of target computation vs epochs (one extra sync)
        self.value_rows = 0                       # critic forward rows issued by the last compute_targets (incl. padding)
        self.last_update_timing = None
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
        # frame index of the newest frame of the acting state at step t (row t), for ppo_gather_stack
        self.k_rows = (torch.arange(T + 1, dtype=torch.int32, device=d) + 3).view(-1, 1).expand(T + 1, N).contiguous()
        self._zero_age = torch.zeros(N, dtype=torch.int32, device=d)
        self.init_pos = torch.tensor(INIT_POS, device=d)
        # the reset frame is a constant of the task: take it from a scratch engine step-free reset
        self.init_frame = self._reset_frame()
        self.frames[:4] = self._encode(self.init_frame) if self.frame_codes else self.init_frame
        self.pos[:4] = self.init_pos
        agent.to(d)
        self.her = None                       # relabelled index records of the current rollout (relabel())
        self.her_out_of_pattern = 0           # records of the last update whose V(s') could not be taken from a neighbour
        # hindsight relabelling looks back over every rollout an episode ending now may have started in:
        # ceil((max_steps - 1) / T) earlier ones; each entry is (pos, term, trunc, reward, age0) of one rollout
        self.max_steps = int(getattr(engine, "max_steps", 50))
        self._hist = collections.deque(maxlen=max(1, -(-(self.max_steps - 1) // self.T)))
        self.her_seed = int(getattr(engine, "seed", 9981))
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
    def _collect_steps(self, uniforms=None, offset_dev=None):
        """The T steps of a rollout: stack gather -> actor -> sample -> engine step -> age.  No host synchronisation
        and no host-dependent control flow, so the same body runs eagerly or under HIP-graph capture."""
        T, N = self.T, self.N
        step_out = [{"obs": None, "matrix": self.frames[t + 4], "pos": self.pos[t + 4], "reward": self.reward[t],
                     "terminated": self.term[t], "truncated": self.trunc[t]} for t in range(T)]
        if self.cache_predictions and self.pred_frames is None:
            self.pred_frames = torch.empty((T, N, 4, 289), dtype=torch.float32, device=self.device)
        self._pred_valid = False
        for t in range(T):
            s4, p4 = ppo_ops.gather_stack(self.frames, self.pos, self.k_rows[t], self.n_all, self.age[t], self.init_frame,
                                          self.init_pos)
            if self.cache_predictions:
                x = self.agent.policy_input(s4)
                self.pred_frames[t] = x[:, 4:]
                a, logp = self.agent.act_batch(s4, p4, self.goal, None if uniforms is None else uniforms[t], x=x)
            elif offset_dev is None:
                a, logp = self.agent.act_batch(s4, p4, self.goal, None if uniforms is None else uniforms[t])
            else:
                a, logp = self.agent.act_batch(s4, p4, self.goal, None, offset_dev=offset_dev, offset_add=t * N)
            self.action[t], self.logp[t] = a, logp
            self.engine.step(a, step_out[t], autoreset=True, policy_idx=True)
            torch.where((self.term[t] | self.trunc[t]) != 0, self._zero_age, self.age[t] + 1, out=self.age[t + 1])
        self._pred_valid = self.cache_predictions

    def _policy_x(self, t_idx, n_idx, after):
        """(network input, positions) of samples (t, n): 4-frame stacks run through agent.policy_input, with the world
        model's predicted frames taken from the rollout's cache for acting states."""
        s, p = self._stacks(t_idx, n_idx, after=after)
        if self._pred_valid and not after:
            return torch.cat([s, self.pred_frames[t_idx.long(), n_idx.long()]], dim=1), p
        return self.agent.policy_input(s), p

    @torch.no_grad()
    def collect(self, uniforms=None):
        """One rollout.  With `use_graph` (small per-GPU batches, where ~25 launches per step make the host the
        bottleneck: 0.5-0.7 ms per step at 256 envs whatever the GPU does) the whole T-step rollout is captured ONCE
        as a HIP graph and replayed: the first rollout runs eagerly (MIOpen searches its kernels then), the second is
        captured, every later one is a single graph launch.  The sampler's stream position is read from device memory
        at replay time, so graph and eager rollouts draw the very same actions."""
        T, N = self.T, self.N
        if not (self.use_graph and uniforms is None and not self.cache_predictions):
            self._collect_steps(uniforms)
        elif not self._graph_warm:
            self._collect_steps(None)
            self._graph_warm = True
        else:
            if self._graph is None:
                from .agent.net.all_net import clear_fold_cache
                torch.cuda.synchronize(self.device)
                g = torch.cuda.CUDAGraph()
                clear_fold_cache(self.agent.trainable_nets())         # nothing folded eagerly may be read by the graph ...
                with torch.cuda.graph(g):
                    self._collect_steps(None, offset_dev=self._graph_base)
                clear_fold_cache(self.agent.trainable_nets())         # ... and nothing folded under capture by eager code
                self._graph = g
            self._graph_base.fill_(self.agent.sample_count)
            self._graph.replay()
            self.agent.sample_count += T * N
        self.env_steps += T * N

    def carry_over(self):
        """Make the last 4 frames the history of the next rollout (and keep what hindsight relabelling needs of
        this one: episodes that end in the next rollout start here)."""
        T = self.T
        self._hist.append((self.pos[4:4 + T].clone(), self.term.clone(), self.trunc.clone(), self.reward.clone(),
                           self.age[0].clone()))
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

    # ------------------------------------------------------------------ hindsight relabelling
    @torch.no_grad()
    def relabel(self, choices=None, max_goals=4):
        """Buffer_gridworld.her_func (env_buffer.py:101-143) for every episode that lies inside this rollout:
        relabelled transitions are index records (t, n, goal', reward', done') produced on the device
        (ppo_her_relabel); update() then trains on the rollout plus these records, like the reference trains on
        its ring buffer with the appended copies.

        Episodes that began in an earlier rollout are relabelled over a window of rollouts long enough to hold any
        episode that ends now (carry_over keeps positions, rewards and done flags of ceil((max_steps-1)/T) earlier
        rollouts); only the records that lie in the current rollout are returned -- the earlier part of such a prefix
        belongs to samples whose frames are gone.

        Agents trained on 9-frame window records (ppo_predictor, self_orinetation_agent: `her_window_delay` = 4) get
        pre_her_func / pre_f_her_func (env_buffer.py:145-280) instead: the goal candidates are the states after steps
        4, 5, ... of the episode (`skip`), see ppo_her_relabel_window in include/twoarmy_ppo.h."""
        T, N = self.T, self.N
        skip = int(getattr(self.agent, "her_window_delay", 0))
        if self.max_steps > HER_MAX_LEN:
            raise ValueError("hindsight relabelling handles episodes of up to %d steps (max_steps = %d)"
                             % (HER_MAX_LEN, self.max_steps))
        start = self.env_steps // N - T                       # global step index of this rollout's first step
        if not self._hist or choices is not None:
            self.her = ppo_ops.her_relabel(self.pos[4:4 + T], self.term, self.trunc, self.age[0].contiguous(),
                                           self.reward, choices, seed=self.her_seed, env_id0=self.engine.env_id0,
                                           step0=start, max_goals=max_goals, skip=skip)
            return self.her
        hist = list(self._hist)
        back = T * len(hist)
        h = ppo_ops.her_relabel(torch.cat([x[0] for x in hist] + [self.pos[4:4 + T]]),
                                torch.cat([x[1] for x in hist] + [self.term]),
                                torch.cat([x[2] for x in hist] + [self.trunc]), hist[0][4],
                                torch.cat([x[3] for x in hist] + [self.reward]), None,
                                seed=self.her_seed, env_id0=self.engine.env_id0, step0=start - back, max_goals=max_goals,
                                skip=skip)
        keep = h["t"] >= back
        self.her = dict(t=(h["t"][keep] - back).contiguous(), n=h["n"][keep].contiguous(), goal=h["goal"][keep].contiguous(),
                        reward=h["reward"][keep].contiguous(), done=h["done"][keep].contiguous())
        self.her["counts"] = torch.bincount(self.her["n"].long(), minlength=N).int()
        return self.her

    def sample_goal(self, t_idx, n_idx, goal2, done):
        """Hook: per-sample goal record carried through targets / update (default: the 2-d goal itself)."""
        return goal2

    def goal_input(self, goal, after):
        """Hook: what the networks get as goal before / after the step, from a sample_goal() record."""
        return goal

    def _values_one(self, t_idx, n_idx, goal, after):
        """critic(s, g) (after=False) or critic(s', g) (after=True) of samples (t, n) with per-sample goals, in chunks."""
        total = t_idx.numel()
        v = torch.empty(total, device=self.device)
        self.agent.critic.eval()
        C = self.value_chunk
        if self.fixed_shapes and total < C:
            C = max(256, 1 << (total - 1).bit_length())          # small sets: next power of two (a handful of shapes in all)
        for i in range(0, total, C):
            sl = torch.arange(i, min(total, i + C), device=self.device)
            n_real = sl.numel()
            if self.fixed_shapes and n_real < C:
                sl = torch.cat([sl, sl[torch.arange(C - n_real, device=self.device) % n_real]])   # pad: one conv batch size
            x, p = self._policy_x(t_idx[sl], n_idx[sl], after)
            self.value_rows += int(sl.numel())
            out = self.agent.critic_value(x, p, self.goal_input(goal[sl], after)).view(-1)
            v[i:i + n_real] = out[:n_real]
        return v

    def _values(self, t_idx, n_idx, goal):
        """critic(s, g) and critic(s', g) of samples (t, n) with per-sample goals."""
        return self._values_one(t_idx, n_idx, goal, False), self._values_one(t_idx, n_idx, goal, True)

    def _values_rollout(self, goal, done):
        """V(s_t) and V(s'_t) of every rollout sample with ONE critic pass over the rollout instead of two: unless the
        episode ends at t, the state after step t IS the acting state of step t + 1 (same frames, same positions, same
        goal input), so V(s'_t) = V(s_{t+1}); only done steps and the last row need their own evaluation
        (the reference evaluates s[:,1:5] and s[:,0:4] separately because its records are independent, PPO.py:112-114)."""
        T, N = self.T, self.N
        idx = torch.arange(T * N, device=self.device)
        t_idx, n_idx = idx // N, idx % N
        v = self._values_one(t_idx, n_idx, goal, False)
        nv = torch.empty_like(v)
        nv.view(T, N)[:-1] = v.view(T, N)[1:]
        need = torch.nonzero((done.view(-1) != 0) | (t_idx == T - 1)).view(-1)
        nv[need] = self._values_one(t_idx[need], n_idx[need], goal[need], True)
        return v, nv

    def _her_next_values(self, h, hgoal, hv):
        """V(s') of relabelled records from V(s) of the record behind them.  Valid only while the records come as runs
        of consecutive steps of one env under one goal, each run ending with its done record (ppo_her_relabel's emission
        order, also after relabel()'s `keep` filter): checked on the device (one sync per update, next to the ones the
        record count already costs); records that break the pattern -- external `choices`, a changed kernel -- get
        their own evaluation of the after-state instead of a neighbour's value."""
        t, n, done = h["t"].long(), h["n"].long(), h["done"] != 0
        g = h["goal"]
        ok_next = torch.zeros_like(done)
        ok_next[:-1] = (n[1:] == n[:-1]) & (t[1:] == t[:-1] + 1) & (g[1:] == g[:-1]).all(-1)
        need = torch.nonzero(done | ~ok_next).view(-1)          # run ends, plus anything out of pattern (incl. the last row)
        hnv = torch.full_like(hv, float("nan"))                  # a row nobody fills would poison the loss visibly
        hnv[:-1] = hv[1:]
        hnv[need] = self._values_one(h["t"][need], h["n"][need], hgoal[need], True)
        self.her_out_of_pattern = int((~done & ~ok_next).sum())
        return hnv

    @torch.no_grad()
    def compute_targets(self):
        T, N = self.T, self.N
        total = T * N
        self.value_rows = 0
        idx = torch.arange(total, device=self.device)
        done = (self.term | self.trunc).contiguous()
        goal = self.sample_goal(idx // N, idx % N, self.goal1.expand(total, 2), done.view(-1))
        v, nv = self._values_rollout(goal, done) if self.reuse_next_values else self._values(idx // N, idx % N, goal)
        adv, target, ret = ppo_ops.gae(self.reward, v.view(T, N), nv.view(T, N), done, gamma=self.agent.gamma,
                                       lam=self.agent.gae_lambda, use_done_mask=self.agent.use_done_mask)
        critic_target = target if self.agent.gae_lambda == 0.0 else ret
        adv, critic_target = adv.view(-1), critic_target.view(-1)
        if self.her is not None and self.her["t"].numel():
            # relabelled records: the reference's one-step target with the relabelled goal and reward (PPO.py:112-114)
            assert self.agent.gae_lambda == 0.0 and not self.agent.use_done_mask, "HER records use the TD(0) targets"
            h = self.her
            hgoal = self.sample_goal(h["t"], h["n"], h["goal"], h["done"])
            if self.reuse_next_values:
                # a relabelled prefix is a run of consecutive steps of one episode under ONE goal that ends with its
                # done record (ppo_her_relabel's emission order), so V(s'_j) of a record that is not done is V(s) of the
                # record behind it; only the prefix ends are evaluated on their after-states
                hv = self._values_one(h["t"], h["n"], hgoal, False)
                hnv = self._her_next_values(h, hgoal, hv)
            else:
                hv, hnv = self._values(h["t"], h["n"], hgoal)
            H = hv.numel()
            hadv, htarget, _ = ppo_ops.gae(h["reward"].view(1, H), hv.view(1, H), hnv.view(1, H), None,
                                           gamma=self.agent.gamma, lam=0.0, use_done_mask=False)
            adv = torch.cat([adv, hadv.view(-1)])
            critic_target = torch.cat([critic_target, htarget.view(-1)])
        if self.agent.normalize_adv:
            if self.agent.grad_sync is not None and torch.distributed.is_initialized() and \
                    torch.distributed.get_world_size() > 1:
                from .. import dist as twdist
                twdist.global_adv_norm_(adv)                      # statistics over every rank's samples
            else:
                ppo_ops.adv_norm_(adv)
        return adv, critic_target

    def update(self, permutations=None):
        ag = self.agent
        T, N = self.T, self.N
        t_begin = time.perf_counter() if self.time_phases else 0.0
        adv, target = self.compute_targets()
        if self.time_phases:
            torch.cuda.synchronize(self.device)
            t_targets = time.perf_counter()
        total = adv.numel()                                           # rollout samples + relabelled records
        base = torch.arange(T * N, device=self.device)
        smp_t, smp_n = (base // N).int(), (base % N).int()
        smp_goal = self.sample_goal(smp_t, smp_n, self.goal1.expand(T * N, 2), (self.term | self.trunc).view(-1))
        if total > T * N:
            h = self.her
            smp_goal = torch.cat([smp_goal, self.sample_goal(h["t"], h["n"], h["goal"], h["done"])])
            smp_t = torch.cat([smp_t, h["t"]])
            smp_n = torch.cat([smp_n, h["n"]])
        flat = smp_t.long() * N + smp_n.long()                        # action / old log-prob are those of (t, n)
        act, logp = self.action.view(-1)[flat], self.logp.view(-1)[flat]
        ag.actor.train(); ag.critic.train()
        la = lv = None
        local_steps = n_steps = -(-total // self.minibatch)
        synced = ag.grad_sync is not None and torch.distributed.is_initialized()
        if synced:
            n_steps = agree_on_steps(n_steps, self.device)
        for ep in range(ag.K_epochs):
            perm = (torch.randperm(total) if permutations is None else torch.as_tensor(permutations[ep])).to(self.device)
            if synced:
                perm = padded_permutation(perm, n_steps, self.minibatch)
            done_steps = 0
            for i in range(0, perm.numel(), self.minibatch):
                idx = perm[i:i + self.minibatch]
                n_valid = None
                if self.fixed_shapes and idx.numel() < self.minibatch <= perm.numel():
                    # the epoch's last, partial minibatch: padded to the full shape with masked rows (same loss and
                    # gradients; a new batch size would cost a MIOpen kernel search -- seconds -- every update)
                    n_valid = idx.numel()
                    idx = torch.cat([idx, idx[torch.arange(self.minibatch - n_valid, device=self.device) % n_valid]])
                with torch.no_grad():
                    x0, p0 = self._policy_x(smp_t[idx], smp_n[idx], False)
                la, lv = ag.minibatch_step_x(x0, p0, self.goal_input(smp_goal[idx], False), act[idx],
                                             logp[idx].view(-1, 1), adv[idx].view(-1, 1), target[idx].view(-1, 1), n_valid)
                done_steps += 1
            assert done_steps == n_steps or not synced, (done_steps, n_steps)
        if ag.use_lr_decay:
            ag.scheduler_actor.step(); ag.scheduler_critic.step()
        if self.time_phases:
            torch.cuda.synchronize(self.device)
            t_end = time.perf_counter()
            self.last_update_timing = {"targets_s": t_targets - t_begin, "epochs_s": t_end - t_targets,
                                       "epoch_s": (t_end - t_targets) / max(1, ag.K_epochs), "samples": int(total)}
        self.her = None
        return la, lv

    def her_switch(self, her, score):
        """The reference's hysteresis (train_ppo.py:128-131): relabelling off above 0.1, on again below 0."""
        return False if score > 0.1 else (True if score < 0.0 else her)

    def running_score(self, score):
        """EMA of episode returns (train_ppo.py:140: score <- 0.99 score + 0.01 ep_reward per finished episode),
        applied once per rollout with the mean return of the E episodes that ended in it."""
        done = (self.term | self.trunc) != 0
        E = int(done.sum())
        if E == 0:
            return score
        mean_ret = float(self.reward.sum()) / E
        k = 0.99 ** E
        return score * k + mean_ret * (1.0 - k)

    def stats(self):
        done = (self.term | self.trunc) != 0
        return {"env_steps": self.env_steps, "episodes": int(done.sum()), "successes": int(self.term.sum()),
                "mean_reward": float(self.reward.mean())}
