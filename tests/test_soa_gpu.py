"""Self-orientation agent on the GPU: update_policy through the HIP target / loss kernels vs the losses the
reference's own update_policy produced (tests/golden/soa.npz), and the batched acting path."""
import numpy as np
import pytest
import torch

from test_soa_cpu import buffer_of, golden, seeded_agent

pytestmark = pytest.mark.gpu


def test_update_policy_losses_match_reference():
    g = golden()
    agent = seeded_agent()
    agent.batch_size, agent.K_epochs = 16, 2
    agent.update_policy(buffer_of(g), "cuda", 0, permutations=g["pol_perms"])
    la = np.array([v for _, v in agent.writer.scalars["loss/action_loss_update"]])
    lv = np.array([v for _, v in agent.writer.scalars["loss/value_loss_update"]])
    assert la.shape == g["pol_action_loss"].shape
    np.testing.assert_allclose(la, g["pol_action_loss"], rtol=0, atol=1e-5)          # north-star bound: 1e-5 fp32
    np.testing.assert_allclose(lv, g["pol_value_loss"], rtol=0, atol=1e-5)


def test_update_orientation_on_gpu_matches_reference():
    g = golden()
    agent = seeded_agent()
    agent.batch_size_pre_agent, agent.K_epochs_pre_agent_position = 16, 2
    agent.update_orientation(buffer_of(g), "cuda", 0, permutations=g["ori_perms"])
    got = np.array([v for _, v in agent.writer.scalars["loss/future_3steps_loss_update"]])
    # first minibatch: same weights, so the loss itself is compared tightly.  Later ones sit behind Adam steps, whose
    # update direction flips with the sign of near-zero gradients, and MIOpen's fp32 backward is not bit-identical to
    # the CPU one the golden came from (the CPU run of this very check, tests/test_soa_cpu.py, holds 1e-5 throughout).
    np.testing.assert_allclose(got[0], g["ori_loss"][0], rtol=0, atol=1e-5)
    np.testing.assert_allclose(got, g["ori_loss"], rtol=0.05, atol=0)


def test_act_batch_soa_semantics():
    g = golden()
    agent = seeded_agent().to("cuda")
    b = buffer_of(g)
    dev = torch.device("cuda")
    s4 = torch.tensor(b["s"][:, :4], dtype=torch.float32, device=dev)
    p4 = torch.tensor(b["p"][:, :4], dtype=torch.float32, device=dev)
    goal = torch.tensor(b["g"], dtype=torch.float32, device=dev)
    u = torch.rand(s4.shape[0], 3, device=dev)
    a, logp, f = agent.act_batch_soa(s4, p4, goal, u)
    # the same draws by hand: inverse-CDF on the orientation heads, then the policy with the extended goal
    with torch.no_grad():
        x8 = agent.policy_input(s4)
        p0, p1 = agent.orient_probs(x8, p4, goal)
        i0 = (torch.cumsum(p0 / p0.sum(1, keepdim=True), 1) > u[:, 0:1]).float().argmax(1)
        i1 = (torch.cumsum(p1 / p1.sum(1, keepdim=True), 1) > u[:, 1:2]).float().argmax(1)
        probs = agent.actor(x8, p4, torch.cat([goal, f], 1))
    assert torch.equal(f[:, 0], i0.float() - 3) and torch.equal(f[:, 1], i1.float() - 3)
    assert bool(((f >= -3) & (f <= 3)).all()) and bool(((a >= 0) & (a < 5)).all())
    want = torch.log(probs.gather(1, a.long().view(-1, 1)).view(-1))
    assert torch.allclose(logp, want, atol=1e-5)
    act, lp, fx, fy = agent.select_action(b["s"][0][:5], b["p"][0][:5], b["g"][0], dev)
    assert 0 <= act < 5 and -3 <= fx <= 3 and -3 <= fy <= 3 and lp <= 0.0


def test_vectorised_soa_trainer_equals_window_records():
    """The index arithmetic of VecSoATrainer (f of the next step, 3-step displacement clamped at the episode end)
    reproduces the reference's 9-frame window records (train_SoA.py:134-183) built literally from the same rollout."""
    from twoarmy_amd.engine import TwoarmyEngine
    from twoarmy_amd.soa.soa_vec import VecSoATrainer
    torch.manual_seed(2)
    N, T = 40, 120
    eng = TwoarmyEngine(6, N, 17, seed=9981)
    agent = seeded_agent()
    tr = VecSoATrainer(agent, eng, rollout_steps=T, minibatch=512, value_chunk=512)
    uni = torch.rand(T + 1, N, 3, device=tr.device)
    tr.collect(uniforms=uni)
    torch.cuda.synchronize()
    frames, pos = tr.frames.cpu().numpy(), tr.pos.cpu().numpy()
    fut, act = tr.future.cpu().numpy(), tr.action.cpu().numpy()
    term, trunc = tr.term.cpu().numpy() != 0, tr.trunc.cpu().numpy() != 0
    init_f, init_p = tr.init_frame.cpu().numpy(), np.array([15.0, 3.0], np.float32)
    # literal 9-deep windows per env, including the four terminal shifts; keyed by the acting step of the record
    rec = {}
    for n in range(N):
        s9 = np.tile(init_f, (9, 1)); p9 = np.tile(init_p, (9, 1)); f5 = np.zeros((5, 2)); k = 0; steps = []
        for t in range(T):
            s9 = np.append(np.delete(s9, 0, 0), [frames[t + 4, n]], 0)
            p9 = np.append(np.delete(p9, 0, 0), [pos[t + 4, n]], 0)
            f5 = np.append(np.delete(f5, 0, 0), [fut[t, n]], 0)
            steps.append(t); k += 1
            if k > 4:
                rec[(steps[-5], n)] = (s9.copy(), p9.copy(), f5.copy())
            if term[t, n] or trunc[t, n]:
                for sh in range(4):
                    s9 = np.append(np.delete(s9, 0, 0), [frames[t + 4, n]], 0)
                    p9 = np.append(np.delete(p9, 0, 0), [pos[t + 4, n]], 0)
                    f5 = np.append(np.delete(f5, 0, 0), [fut[t, n]], 0)
                    if k + sh + 1 > 4:
                        rec[(steps[-4 + sh] if len(steps) >= 4 - sh else None, n)] = (s9.copy(), p9.copy(), f5.copy())
                s9 = np.tile(init_f, (9, 1)); p9 = np.tile(init_p, (9, 1)); f5 = np.zeros((5, 2)); k = 0; steps = []
    rec.pop((None, 0), None)
    keys = [k for k in rec if k[0] is not None]
    assert len(keys) > 1000
    t_idx = torch.tensor([k[0] for k in keys], device=tr.device)
    n_idx = torch.tensor([k[1] for k in keys], device=tr.device)
    s0, p0 = tr._stacks(t_idx.int(), n_idx.int(), after=False)
    s1, p1 = tr._stacks(t_idx.int(), n_idx.int(), after=True)
    done = ((tr.term | tr.trunc) != 0)[t_idx, n_idx]
    g6 = tr.sample_goal(t_idx, n_idx, tr.goal1.expand(len(keys), 2), done)
    S = np.stack([rec[k][0] for k in keys]); P = np.stack([rec[k][1] for k in keys]); F = np.stack([rec[k][2] for k in keys])
    assert np.array_equal(s0.cpu().numpy(), S[:, 0:4]) and np.array_equal(s1.cpu().numpy(), S[:, 1:5])
    assert np.array_equal(p0.cpu().numpy(), P[:, 0:4]) and np.array_equal(p1.cpu().numpy(), P[:, 1:5])
    assert np.array_equal(g6[:, 2:4].cpu().numpy(), F[:, 0]) and np.array_equal(g6[:, 4:6].cpu().numpy(), F[:, 1])
    # orientation targets of successful episodes == p[:,6] - p[:,3] of their window records (the seeded policy never
    # reaches the goal in 120 steps, so every finished episode is declared a success for this check)
    real_term = tr.term.clone()
    tr.term.copy_(tr.term | tr.trunc)
    ot, on, og, disp = tr.orientation_samples()
    tr.term.copy_(real_term)
    assert ot.numel() > 500 and float(disp.abs().max()) <= 3
    for t, n, d in zip(ot.tolist(), on.tolist(), disp.cpu().numpy()):
        w = rec[(t, n)][1]
        assert np.array_equal(d, w[6] - w[3]), (t, n)
    # hindsight records of the unsuccessful episodes: displacement clamped at the relabelled end of each prefix
    h = tr.relabel()
    ot, on, og, disp = tr.orientation_samples()
    ht, hn, hd, hg = (h[k].cpu().numpy() for k in ("t", "n", "done", "goal"))
    assert ot.numel() == ht.size > 0 and np.array_equal(og.cpu().numpy(), hg)
    age = tr.age.cpu().numpy()

    def state_pos(sidx, n):                     # position of the state before step sidx
        return init_p if age[sidx, n] == 0 else pos[sidx + 3, n]
    ends = np.flatnonzero(hd)
    disp_h = disp.cpu().numpy()
    for i, (t, n) in enumerate(zip(ht, hn)):
        u = ht[ends[np.searchsorted(ends, i)]]
        want = pos[min(t + 3, u + 1) + 3, n] - state_pos(t, n)
        assert np.array_equal(disp_h[i], want), (i, t, n, u)
    # and the two learners run end to end (policy, then orientation), with hindsight records
    agent.K_epochs, agent.K_epochs_pre_agent_position = 1, 1
    # whole minibatches only: every distinct batch size costs a MIOpen kernel search
    total, n_or = T * N + int(h["t"].numel()), int(ot.numel())
    la, lv = tr.update(permutations=[torch.randperm(total)[:total // 512 * 512]],
                       orient_permutations=[torch.randperm(n_or)[:n_or // 512 * 512]])
    assert np.isfinite(float(la)) and np.isfinite(float(lv)) and np.isfinite(float(tr.last_orientation_loss))
    pending = tr.pending_future.clone()
    tr.carry_over()
    tr.collect()
    assert torch.equal(tr.future[0], pending)          # the draw made for the state after the last step is the one used
    eng.close()


def test_train_soa_entry_point_smoke():
    from twoarmy_amd.soa import train_SoA
    tr = train_SoA.main(["--env", "MiniGrid-twoarmy-17x17-v4", "--num_envs", "32", "--rollout_steps", "60",
                         "--minibatch", "256", "--updates", "2", "--k_epochs", "1", "--k_epochs_orientation", "1",
                         "--her", "False", "--cuda", "cuda:0"])
    assert tr.env_steps == 2 * 60 * 32 and tr.agent.agent_position_preditor.Px.out_features == 7


def test_success_replay_equals_literal_fifo_of_99_episodes():
    """The orientation head's success replay across updates (train_SoA.py:200-205: fp_terminate_buffer, FIFO of the
    last 99 goal-reaching episodes, never cleared by the buffer reset) vs a literal construction: synthetic rollouts
    with random episode structure are written into the trainer; after every "update" the trained sample set
    (current goal-reaching episodes + replay) must equal, as a multiset of (displacement, acting position) rows, the
    literal FIFO's content -- and, when a rollout alone holds more than 99 successes, every one of them."""
    import collections
    from twoarmy_amd.engine import TwoarmyEngine
    from twoarmy_amd.soa.soa_vec import VecSoATrainer
    rng = np.random.RandomState(7)
    init_p = np.array([15.0, 3.0], np.float32)

    def run(N, T, n_rollouts, p_term):
        eng = TwoarmyEngine(6, N, 17, seed=9981)
        tr = VecSoATrainer(seeded_agent(), eng, rollout_steps=T, minibatch=512, value_chunk=512)
        fifo = collections.deque()
        age0 = np.zeros(N, np.int64)
        seen_big = False
        for r in range(n_rollouts):
            term = np.zeros((T, N), np.uint8); trunc = np.zeros((T, N), np.uint8)
            pos = rng.randint(1, 16, size=(T + 4, N, 2)).astype(np.float32)
            pos[:4] = tr.pos[:4].cpu().numpy() if r else init_p
            age = np.zeros((T + 1, N), np.int64); age[0] = age0
            for n in range(N):
                t = 0
                left = rng.randint(4, 18) - min(int(age0[n]), 3)
                for t in range(T):
                    left -= 1
                    if left <= 0:
                        (term if rng.rand() < p_term else trunc)[t, n] = 1
                        left = rng.randint(4, 18)
                    age[t + 1, n] = 0 if (term[t, n] or trunc[t, n]) else age[t, n] + 1
            tr.term.copy_(torch.tensor(term)); tr.trunc.copy_(torch.tensor(trunc))
            tr.pos.copy_(torch.tensor(pos)); tr.age.copy_(torch.tensor(age, dtype=torch.int32))
            tr.her = None
            # --- literal: goal-reaching episodes of this rollout in completion order
            eps = []
            for n in range(N):
                start = 0
                for t in range(T):
                    if term[t, n] or trunc[t, n]:
                        if term[t, n]:
                            rows = []
                            for s in range(start, t + 1):
                                p_now = init_p if age[s, n] == 0 else pos[s + 3, n]
                                p_fut = pos[min(s + 3, t + 1) + 3, n]
                                rows.append(tuple(np.concatenate([p_fut - p_now, p_now]).tolist()))
                            eps.append(((t, n), rows))
                        start = t + 1
            eps.sort(key=lambda e: e[0])
            n_cur = len(eps)
            for e in eps:
                fifo.append(e[1])
                if len(fifo) > 99:
                    fifo.popleft()
            want = [row for e in (fifo if n_cur <= 99 else [x[1] for x in eps]) for row in e]
            # --- trainer
            t_idx, n_idx, goal2, disp = tr.orientation_samples()
            assert tr._n_success_samples == sum(len(e[1]) for e in eps)
            rep = tr.replay_samples(n_cur)
            s0, p0 = tr._stacks(t_idx, n_idx, after=False)
            got = torch.cat([disp, p0[:, 3]], 1).cpu().numpy().tolist()
            if rep is not None:
                got += torch.cat([rep["disp"], rep["p0"][:, 3]], 1).cpu().numpy().tolist()
                assert bool((rep["goal2"] == tr.goal1).all())
            assert sorted(map(tuple, got)) == sorted(want), (r, n_cur, len(got), len(want))
            seen_big |= n_cur > 99
            tr.remember_successes(t_idx, n_idx, goal2, disp)
            assert len(tr._replay) == min(99, len(fifo)) and len(tr._replay) <= 99
            tr.carry_over()
            age0 = age[T]
        eng.close()
        return seen_big

    assert not run(N=6, T=60, n_rollouts=6, p_term=0.5)          # < 99 successes per rollout: the FIFO spans several updates
    assert run(N=64, T=60, n_rollouts=2, p_term=0.9)             # > 99 successes in one rollout: all of them train
