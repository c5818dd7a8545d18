"""Facade env, VecEnv and the vectorised rollout storage on the GPU."""
import numpy as np
import pytest
import torch

from golden_util import load_traces

pytestmark = pytest.mark.gpu
TRACES, SEED = load_traces()


@pytest.mark.parametrize("idx", [0, 3, 4, 6, 12, 17, 24, 30, 38])
def test_facade_env_replays_reference_traces(idx):
    """gym_minigrid facade (MiniGridEnv API) + Env_transact on recorded reference traces."""
    from twoarmy_amd.gym_minigrid import make
    from twoarmy_amd.soa.env_buffer import Env_transact
    tr = TRACES[idx]
    if int(tr["natural"]):
        pytest.skip("natural-stream traces need explicit draws (covered by test_engine_gpu)")
    env = make("MiniGrid-twoarmy-17x17-v%d" % int(tr["variant"]), seed=SEED, env_id=int(tr["env_id"]), tile_size=17)
    et = Env_transact()
    errs = {1: AttributeError, 2: AssertionError, 3: TypeError}
    for k, op in enumerate(tr["op"]):
        ctx = "%s op#%d=%d" % (tr["name"], k, op)
        if op == -1:
            obs = env.reset()
            assert np.array_equal(obs["image"], tr["obs"][k]), ctx
        elif int(tr["err"][k]):
            with pytest.raises(errs[int(tr["err"][k])]):
                env.step(int(op))
        else:
            obs, r, te, trn, info = env.step(int(op))
            assert np.array_equal(obs["image"], tr["obs"][k]) and obs["direction"] == 3, ctx
            assert r == float(tr["reward"][k]) and te == bool(tr["term"][k]) and trn == bool(tr["trunc"][k]), ctx
        assert env.agent_pos == tuple(tr["agent"][k]) and env.step_count == int(tr["scal"][k][0]), ctx
        assert np.array_equal(env.grid.encode(), tr["grid"][k]), ctx
        assert np.array_equal(et.matrix_env(env), tr["matrix"][k]), ctx
        a, g = et.data_env(env)
        assert np.concatenate([a, g]).tolist() == tr["pos"][k].tolist(), ctx
        assert [o.cur_pos for o in env.obstacles] == [tuple(p) for p in tr["balls"][k]], ctx
        if k % 16 == 0:       # window / auxiliary views of the same state (env_buffer.py:336-362, 430-437)
            sm9, st9 = et.predata_reset(env)
            assert sm9.shape == (9, 289) and np.array_equal(sm9[8], tr["matrix"][k]) and st9[0].tolist() == a.tolist()
            m, m8 = et.pre_col(env)
            assert np.array_equal(m, tr["matrix"][k]) and m8.shape == (8, 289)
            st, st10 = et.free_env(env)
            bx = int(tr["balls"][k][0][0])
            span = {6: [8, 9, 8, 10], 7: [8, 6, 8, 10]}.get(bx, [8, 6, 8, 7])
            assert st.tolist() == a.tolist() + span + g.tolist() and st10.shape == (10, 8)
    assert env.action_space.n == 7 and env.actions.done == 6 and et.env_action(env, 4) == 6
    env.close()


def test_view_size_wrapper_reslices_the_state_after_the_step():
    """ViewSizeWrapper(env, 7) (wrappers.py:428-460) calls gen_obs_grid(7) + encode AFTER env.step() has returned, i.e. on
    the state after this step's wall drop / patrol spawn, whereas the observation step() itself returns was taken before
    them (twoarmy_v6.py:182-198 run after MiniGridEnv.step's gen_obs).  So the wrapper's image == the 7x7 view of the
    CURRENT state (an env built with agent_view_size=7, whose views are pinned by the reference's goldens in
    tests/test_engine_gpu.py), and differs from that env's step observation exactly at the drop step."""
    from twoarmy_amd.gym_minigrid import make
    from twoarmy_amd.gym_minigrid.wrappers import ViewSizeWrapper
    acts = [2, 2, 1, 1, 2, 2, 2, 0, 3, 1, 2, 2, 2, 2, 1, 2, 6, 2, 2, 2, 0, 0, 2, 2, 2, 1, 1, 1, 2, 2]
    base = make("MiniGrid-twoarmy-17x17-v4", seed=SEED, env_id=3, tile_size=17)
    small = make("MiniGrid-twoarmy-17x17-v4", seed=SEED, env_id=3, tile_size=17, agent_view_size=7)
    env = ViewSizeWrapper(base, agent_view_size=7)
    assert env.observation_space["image"].shape == (7, 7, 3) and env.action_space.n == 7 and env.agent_pos == base.agent_pos
    o, o7 = env.reset(), small.reset()
    assert o["image"].shape == (7, 7, 3) and np.array_equal(o["image"], o7["image"]) and o["direction"] == 3
    differs = []
    for k, a in enumerate(acts):
        pone_before, patrol_before = small.pone, small.patrol
        (o, r, te, tr, _), (o7, r7, te7, tr7, _) = env.step(a), small.step(a)
        assert (r, te, tr) == (r7, te7, tr7) and env.agent_pos == small.agent_pos, k
        assert np.array_equal(o["image"], small.gen_obs()["image"]), k          # the view of the state after the step
        assert o["image"][3, 6].tolist() == [1, 0, 0]                           # the agent's own cell encodes as empty
        if not np.array_equal(o["image"], o7["image"]):
            differs.append(k)
            assert (small.pone and not pone_before) or (small.patrol and not patrol_before)   # only a drop / spawn step
        if te or tr:
            break
    assert k >= 10 and len(differs) <= 2
    with pytest.raises(AssertionError):
        ViewSizeWrapper(base, agent_view_size=4)
    base.close(); small.close()


def test_vecenv_autoreset_semantics():
    from twoarmy_amd.vecenv import TwoarmyVecEnv
    env = TwoarmyVecEnv("MiniGrid-twoarmy-17x17-v6", num_envs=256, seed=SEED)
    obs0 = env.reset().clone()
    assert obs0.shape == (256, 17, 17, 3) and bool((obs0 == obs0[0]).all())
    saw_done = 0
    for t in range(60):
        a = torch.full((256,), 2 if t % 3 else 1, dtype=torch.int64, device=env.device)     # right / up
        obs, r, te, tr, info = env.step(a)
        done = te | tr
        if bool(done.any()):
            saw_done += int(done.sum())
            assert bool((obs[done] == obs0[0]).all())                  # next-episode observation
            assert not bool((info["final_observation"][done] == obs0[0]).all())
        assert env.state_matrix.shape == (256, 289) and env.agent_yx.shape == (256, 2)
    assert saw_done > 0
    env.close()


def test_rollout_stacks_equal_reference_style_stacks():
    """The on-the-fly 4-frame stacks (frames + age + ppo_gather_stack) equal the literal 5-deep
    np.delete/np.append stacks of soa/train_ppo.py:104-121 for every (t, n)."""
    from twoarmy_amd.engine import TwoarmyEngine
    from twoarmy_amd.soa.agent.PPO import PPO
    from twoarmy_amd.soa.ppo_vec import VecPPOTrainer
    torch.manual_seed(0)
    N, T = 48, 70
    eng = TwoarmyEngine(6, N, 17, seed=SEED)
    agent = PPO()
    tr = VecPPOTrainer(agent, eng, rollout_steps=T, minibatch=256)
    uni = torch.rand(T, N, device=tr.device)
    tr.collect(uniforms=uni)
    torch.cuda.synchronize()
    frames = tr.frames.cpu().numpy(); pos = tr.pos.cpu().numpy()
    done = ((tr.term | tr.trunc) != 0).cpu().numpy()
    init_f, init_p = tr.init_frame.cpu().numpy(), np.array([15.0, 3.0], np.float32)
    # literal stacks, one env at a time
    lit_s = np.zeros((T, N, 5, 289), np.float32); lit_p = np.zeros((T, N, 5, 2), np.float32)
    for n in range(N):
        s = np.tile(init_f, (5, 1)); p = np.tile(init_p, (5, 1))
        for t in range(T):
            s = np.append(np.delete(s, 0, 0), [frames[t + 4, n]], 0)
            p = np.append(np.delete(p, 0, 0), [pos[t + 4, n]], 0)
            lit_s[t, n], lit_p[t, n] = s, p
            if done[t, n]:
                s = np.tile(init_f, (5, 1)); p = np.tile(init_p, (5, 1))
    idx = torch.arange(T * N, device=tr.device)
    s0, p0 = tr._stacks(idx // N, idx % N, after=False)
    s1, p1 = tr._stacks(idx // N, idx % N, after=True)
    assert np.array_equal(s0.cpu().numpy().reshape(T, N, 4, 289), lit_s[:, :, 0:4])
    assert np.array_equal(s1.cpu().numpy().reshape(T, N, 4, 289), lit_s[:, :, 1:5])
    assert np.array_equal(p0.cpu().numpy().reshape(T, N, 4, 2), lit_p[:, :, 0:4])
    assert np.array_equal(p1.cpu().numpy().reshape(T, N, 4, 2), lit_p[:, :, 1:5])
    assert done.sum() > 0
    # and one optimisation pass runs end to end
    agent.K_epochs = 1
    la, lv = tr.update()
    assert np.isfinite(float(la)) and np.isfinite(float(lv))
    tr.carry_over()
    tr.collect()
    eng.close()


def test_frame_codes_trainer_inputs_bit_identical_losses_1e5():
    """Storing the rollout as uint8 code frames (ppo_gather_stack_u8): the actions and every policy input are
    BIT-identical to the float-frame trainer's; the update losses agree within 1e-5 (MIOpen's conv backward
    accumulates in a run-dependent order, so two runs on identical inputs are themselves only equal to ~2e-6)."""
    from twoarmy_amd.engine import TwoarmyEngine
    from twoarmy_amd.soa.agent.PPO import PPO
    from twoarmy_amd.soa.ppo_vec import VecPPOTrainer
    N, T = 64, 40
    res = []
    for codes in (False, True):
        torch.manual_seed(3)
        eng = TwoarmyEngine(4, N, 17, seed=SEED)
        agent = PPO()
        agent.K_epochs = 1
        tr = VecPPOTrainer(agent, eng, rollout_steps=T, minibatch=512, frame_codes=codes)
        g = torch.Generator(device="cpu").manual_seed(5)
        uni = torch.rand(T, N, generator=g).to(tr.device)
        tr.collect(uniforms=uni)
        idx = torch.arange(T * N, device=tr.device)
        s1, p1 = tr._stacks(idx // N, idx % N, after=True)
        perms = [torch.randperm(T * N, generator=g).to(tr.device)]
        la, lv = tr.update(permutations=perms)
        res.append((tr.action.clone(), s1.clone(), p1.clone(), float(la), float(lv)))
        assert tr.frames.dtype == (torch.uint8 if codes else torch.float32)
        eng.close()
    a, b = res
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    # identical inputs to both updates; MIOpen's conv backward accumulates in a run-dependent order (1.6e-6 seen)
    assert abs(a[3] - b[3]) <= 1e-5 and abs(a[4] - b[4]) <= 1e-5


def test_trainer_with_device_her_records():
    """relabel() + update(): the relabelled index records extend the sample set; their TD(0) targets use the
    relabelled goal and reward (checked against a direct torch evaluation of PPO.py:112-114)."""
    from twoarmy_amd.engine import TwoarmyEngine
    from twoarmy_amd.soa.agent.PPO import PPO
    from twoarmy_amd.soa.ppo_vec import VecPPOTrainer
    torch.manual_seed(1)
    N, T = 96, 120
    eng = TwoarmyEngine(4, N, 17, seed=SEED)
    agent = PPO()
    agent.K_epochs = 1
    tr = VecPPOTrainer(agent, eng, rollout_steps=T, minibatch=2048, value_chunk=2048)
    tr.collect()
    h = tr.relabel()
    H = int(h["t"].numel())
    assert H > 0 and int(h["counts"].sum()) == H
    # every record points at a real (t, n) of an episode and carries an achieved position of that episode as goal
    t, n = h["t"].long(), h["n"].long()
    ends = torch.nonzero(h["done"]).view(-1)
    assert bool((h["reward"][ends] == 0.9).all())
    assert bool((tr.pos[4:][t[ends], n[ends]] == h["goal"][ends]).all())
    adv, target = tr.compute_targets()
    assert adv.numel() == T * N + H
    with torch.no_grad():
        s0, p0 = tr._stacks(h["t"], h["n"], after=False)
        s1, p1 = tr._stacks(h["t"], h["n"], after=True)
        agent.critic.eval()
        v = agent.critic(agent.policy_input(s0), p0, h["goal"]).view(-1)
        nv = agent.critic(agent.policy_input(s1), p1, h["goal"]).view(-1)
        want_t = h["reward"] + agent.gamma * nv
    assert torch.allclose(target[T * N:], want_t, atol=1e-5) and torch.allclose(adv[T * N:], want_t - v, atol=1e-5)
    assert tr.her_out_of_pattern == 0              # the kernel emits runs of one episode under one goal: V(s') came from neighbours
    # records in any other order (external `choices`, a changed kernel): the shortcut notices and evaluates their after-states
    perm = torch.randperm(H, device=h["t"].device)
    tr.her = {k: (val[perm].contiguous() if k != "counts" else val) for k, val in h.items()}
    adv2, target2 = tr.compute_targets()
    assert tr.her_out_of_pattern > 0
    assert torch.allclose(target2[T * N:], want_t[perm], atol=1e-5) and torch.allclose(adv2[T * N:], (want_t - v)[perm], atol=1e-5)
    tr.her = h
    total = T * N + H                           # whole minibatches only: every distinct batch size costs a MIOpen search
    la, lv = tr.update(permutations=[torch.randperm(total)[:total // 2048 * 2048]])
    assert np.isfinite(float(la)) and np.isfinite(float(lv)) and tr.her is None
    assert tr.her_switch(True, 0.2) is False and tr.her_switch(False, -0.1) is True and tr.her_switch(False, 0.05) is False
    eng.close()


@pytest.mark.parametrize("window_records", [False, True])
def test_hindsight_records_across_the_rollout_boundary(window_records):
    """Episodes that start in one rollout and end in the next: relabel() works on the two-rollout window and returns
    the records that lie in the current rollout -- equal to the oracle run over the concatenated rollouts.  With the
    predictor agent the relabelling is the window-record variant (pre_her_func: goal candidates from the fifth step)."""
    import her_oracle
    from twoarmy_amd.engine import TwoarmyEngine
    from twoarmy_amd.soa.agent.PPO import PPO
    from twoarmy_amd.soa.agent.PPO_Predictor import ppo_predictor
    from twoarmy_amd.soa.ppo_vec import VecPPOTrainer
    torch.manual_seed(4)
    N, T = 48, 40                                   # 50-step episodes: every second episode straddles a boundary
    skip = 4 if window_records else 0
    eng = TwoarmyEngine(4, N, 17, seed=SEED)
    tr = VecPPOTrainer(ppo_predictor() if window_records else PPO(), eng, rollout_steps=T, minibatch=512)
    tr.collect()
    first = {k: v.clone() for k, v in dict(pos=tr.pos[4:], term=tr.term, trunc=tr.trunc, reward=tr.reward, age0=tr.age[0]).items()}
    h1 = tr.relabel()
    tr.her = None
    tr.carry_over()
    tr.collect()
    h2 = tr.relabel()
    cat = lambda a, b: np.concatenate([a.cpu().numpy(), b.cpu().numpy()])          # noqa: E731
    want = her_oracle.relabel(cat(first["pos"], tr.pos[4:]), cat(first["term"], tr.term), cat(first["trunc"], tr.trunc),
                              first["age0"].cpu().numpy(), cat(first["reward"], tr.reward), seed=tr.her_seed,
                              env_id0=0, step0=0, skip=skip)
    keep = want["t"] >= T
    assert keep.sum() > 0 and (want["t"][keep] - T < 10).sum() > 0       # records right after the boundary: straddling episodes
    for k in ("n", "goal", "reward", "done"):
        assert np.array_equal(h2[k].cpu().numpy(), want[k][keep]), k
    assert np.array_equal(h2["t"].cpu().numpy(), want["t"][keep] - T)
    # the first rollout on its own was relabelled with the same picks (same global step keys)
    w1 = her_oracle.relabel(first["pos"].cpu().numpy(), first["term"].cpu().numpy(), first["trunc"].cpu().numpy(),
                            first["age0"].cpu().numpy(), first["reward"].cpu().numpy(), seed=tr.her_seed, env_id0=0, step0=0,
                            skip=skip)
    assert np.array_equal(h1["t"].cpu().numpy(), w1["t"]) and np.array_equal(h1["goal"].cpu().numpy(), w1["goal"])
    eng.close()


def test_train_ppo_entry_point_smoke():
    from twoarmy_amd.soa import train_ppo
    tr = train_ppo.main(["--env", "MiniGrid-twoarmy-17x17-v4", "--num_envs", "64", "--rollout_steps", "16",
                         "--minibatch", "256", "--updates", "2", "--k_epochs", "1", "--her", "False", "--cuda", "cuda:0"])
    assert tr.env_steps == 2 * 16 * 64


def test_train_ppo_predictor_entry_point_smoke():
    from twoarmy_amd.soa import train_ppo_predictor
    tr = train_ppo_predictor.main(["--env", "MiniGrid-twoarmy-17x17-v4", "--num_envs", "32", "--rollout_steps", "8",
                                   "--minibatch", "128", "--updates", "1", "--k_epochs", "1", "--her", "False", "--cuda", "cuda:0"])
    assert tr.env_steps == 8 * 32 and tr.agent.actor.bone1.cnn_base[0].weight.shape[1] == 8


def test_offline_world_model_pipeline_end_to_end(tmp_path):
    """f4: datacol_predictor (windows from the HIP engine) -> train_encoder_decoder -> train_predictor -> the
    checkpoint loads into the PPO+predictor agent (keys model_encoder / model_decoder / model_predictor)."""
    import glob
    from twoarmy_amd.soa import datacol_predictor, train_encoder_decoder, train_predictor
    from twoarmy_amd.soa.agent.PPO_Predictor import ppo_predictor
    data = datacol_predictor.main(["--env", "MiniGrid-twoarmy-17x17-v4", "--buffer_pre_capacity", "288", "--num_envs", "16",
                                   "--rollout_steps", "64", "--log_dir", str(tmp_path / "data")])
    buf = np.load(data)
    assert buf.shape == (288,) and buf["s"].shape == (288, 9, 289) and set(np.unique(buf["s"])) <= {-0.9, -0.5, 0.3, 0.9}
    m = train_encoder_decoder.main(["--buffer_file", data, "--num_episodes", "2", "--batch_size", "32",
                                    "--log_dir", str(tmp_path / "runs")])
    tr = [v for _, v in m.en_de_writer.scalars["loss/en_de_train_loss_update"]]
    assert len(tr) == 2 * 9 and tr[-1] < tr[0]                                  # 259 train frames / 32, loss goes down
    ck = glob.glob(str(tmp_path / "runs" / "param" / "ppo_encoder_decoder" / "*" / "*.pkl"))
    assert len(ck) == 1
    p = train_predictor.main(["--buffer_file", data, "--net_file", ck[0], "--num_episodes", "3", "--batch_size", "32",
                              "--log_dir", str(tmp_path / "runs2")])
    assert all(not q.requires_grad for q in p.encoder.parameters())
    ck2 = glob.glob(str(tmp_path / "runs2" / "param" / "ppo_encoder_decoder_predictor" / "*" / "*.pkl"))
    assert len(ck2) == 1
    agent = ppo_predictor()
    agent.load_world_model(torch.load(ck2[0], map_location="cpu", weights_only=True))
    assert torch.equal(agent.predictor.state_dict()["recurrent_model.weight_ih_l0"],
                       p.predictor.state_dict()["recurrent_model.weight_ih_l0"].cpu())


def test_full_size_config2_collect_replay_and_minibatch():
    """BASELINE configs[2] at FULL size (4096 envs x 128 steps, v6) through VecPPOTrainer.collect with the actor in
    the loop, then size-independent checks:
      * replaying the recorded actions through ONE pipelined tw_rollout launch from a fresh engine reproduces every
        frame / position / reward / done flag of the per-step collection bit for bit (record layout, fast kernel);
      * rewards take only the reference's five values, age counts steps since reset;
      * one PPO minibatch (8192 samples, both networks; the configs[2] minibatch of 32768 only makes MIOpen's first-use
        kernel search four times longer, it checks nothing more): the fused HIP loss equals the numpy oracle's loss
        on the same probabilities / values within 1e-5."""
    import ppo_oracle as po
    from twoarmy_amd import ppo_ops
    from twoarmy_amd.engine import TwoarmyEngine
    from twoarmy_amd.soa.agent.PPO import PPO
    from twoarmy_amd.soa.ppo_vec import VecPPOTrainer
    N, T, MB = 4096, 128, 8192
    torch.manual_seed(9981)
    eng = TwoarmyEngine(6, N, 17, seed=SEED)
    agent = PPO()
    agent.K_epochs = 1
    tr = VecPPOTrainer(agent, eng, rollout_steps=T, minibatch=MB)
    tr.collect()
    torch.cuda.synchronize()
    # --- replay through the pipelined kernel
    eng2 = TwoarmyEngine(6, N, 17, seed=SEED)
    out = eng2.alloc_outputs(T)
    eng2.rollout(T, out, actions=tr.action.contiguous(), autoreset=True, policy_idx=True)
    torch.cuda.synchronize()
    assert torch.equal(out["matrix"], tr.frames[4:4 + T])
    assert torch.equal(out["pos"], tr.pos[4:4 + T])
    assert torch.equal(out["reward"], tr.reward) and torch.equal(out["terminated"], tr.term)
    assert torch.equal(out["truncated"], tr.trunc)
    a, b = eng.get_state(), eng2.get_state()
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    # --- properties
    vals = torch.tensor([-0.01, -0.1, -0.9, 0.2, 0.9], device=tr.device)
    assert bool((tr.reward.view(-1, 1) == vals.view(1, -1)).any(1).all())
    done = (tr.term | tr.trunc) != 0
    assert int(done.sum()) > N                                   # every env finished at least one episode on average
    age = tr.age.cpu().numpy(); d = done.cpu().numpy()
    assert np.array_equal(age[1:], np.where(d, 0, age[:-1] + 1)) and age.max() < 50
    # --- one full-size minibatch
    idx = torch.randperm(T * N, device=tr.device)[:MB]
    t_idx, n_idx = (idx // N).int(), (idx % N).int()
    s0, p0 = tr._stacks(t_idx, n_idx, after=False)
    goal = tr.goal1.expand(MB, 2).contiguous()
    agent.actor.train(); agent.critic.train()
    with torch.no_grad():
        probs = agent.actor_probs(s0, p0, goal)
        value = agent.critic_value(s0, p0, goal)
    act = tr.action.view(-1)[idx.long()]
    logp = tr.logp.view(-1)[idx.long()].view(-1, 1)
    adv = torch.randn(MB, 1, device=tr.device) * 0.1
    tgt = torch.randn(MB, 1, device=tr.device) * 0.1
    la, lv = agent.minibatch_step(s0, p0, goal, act, logp, adv, tgt)
    q, logits, ent = po.categorical(probs.cpu().numpy())
    lp = logits[np.arange(MB), act.cpu().numpy()].reshape(-1, 1)
    ratio = np.exp(lp - logp.cpu().numpy())
    advn = adv.cpu().numpy()
    want_la = float(np.mean(-np.minimum(ratio * advn, np.clip(ratio, 0.9, 1.1) * advn) - 0.01 * ent.reshape(-1, 1), dtype=np.float64))
    dv = value.cpu().numpy() - tgt.cpu().numpy()
    want_lv = float(np.mean(np.where(np.abs(dv) < 1, 0.5 * dv * dv, np.abs(dv) - 0.5), dtype=np.float64))
    assert abs(float(la) - want_la) < 1e-5 and abs(float(lv) - want_lv) < 1e-5
    eng.close(); eng2.close()


def test_fixed_shape_padding_changes_nothing():
    """VecPPOTrainer pads the epoch's last minibatch and the last value chunk to the full shape (masked rows) so that
    every conv launch of an update has one batch size: targets and losses agree within 1e-5 (a different batch size
    makes MIOpen pick a different kernel, i.e. a different fp32 summation order; the masked loss kernel itself is exact,
    tests/test_ppo_gpu.py::test_masked_fixed_shape_loss_equals_unpadded)."""
    from twoarmy_amd.engine import TwoarmyEngine
    from twoarmy_amd.soa.agent.PPO import PPO
    from twoarmy_amd.soa.ppo_vec import VecPPOTrainer
    N, T = 48, 50
    out = []
    for fixed in (False, True):
        torch.manual_seed(11)
        eng = TwoarmyEngine(4, N, 17, seed=SEED)
        agent = PPO()
        agent.K_epochs = 1
        tr = VecPPOTrainer(agent, eng, rollout_steps=T, minibatch=1024, value_chunk=1024)
        tr.fixed_shapes = fixed
        g = torch.Generator(device="cpu").manual_seed(3)
        tr.collect(uniforms=torch.rand(T, N, generator=g).to(tr.device))
        adv, target = tr.compute_targets()
        la, lv = tr.update(permutations=[torch.randperm(T * N, generator=g)])
        out.append((adv.clone(), target.clone(), float(la), float(lv)))
        eng.close()
    (a0, t0, la0, lv0), (a1, t1, la1, lv1) = out
    assert torch.allclose(a0, a1, atol=2e-5) and torch.allclose(t0, t1, atol=2e-5)
    assert abs(la0 - la1) < 1e-5 and abs(lv0 - lv1) < 1e-5


def test_what_fp16_frames_would_cost_vs_code_frames():
    """BASELINE configs[4] says "fp16 obs encode".  Recorded here: an fp16 frame plane is NOT exact (0.9 -> 0.89990234,
    0.3 -> 0.30004883; only -0.5 survives), it moves the critic's value by more than the 1e-5 loss tolerance, while the
    uint8 code plane (TW_F_MATRIX_CODE, half the bytes of fp16) expands to the very same fp32 inputs."""
    from twoarmy_amd.engine import TwoarmyEngine
    from twoarmy_amd.soa.agent.PPO import PPO
    torch.manual_seed(4)
    eng = TwoarmyEngine(6, 256, 17, seed=SEED)
    acts = eng.fill_actions(40)
    of, oc = eng.alloc_outputs(40), eng.alloc_outputs(40, matrix_codes=True)
    st = eng.get_state()
    eng.rollout(40, of, actions=acts)
    eng.set_state(*st)
    eng.rollout(40, oc, actions=acts)
    frames = of["matrix"][-4:].permute(1, 0, 2).contiguous()                        # [N, 4, 289] exact fp32 frames
    assert torch.equal(TwoarmyEngine.decode_matrix(oc["matrix"][-4:]).permute(1, 0, 2), frames)     # codes: exact
    f16 = frames.half().float()
    err = (f16 - frames).abs().max().item()
    assert 9.0e-5 < err < 1.0e-4                                                    # fp16(0.9) is off by 9.8e-5
    agent = PPO().to("cuda:0")
    agent.critic.eval()
    pos = of["pos"][-4:].permute(1, 0, 2).contiguous()
    goal = torch.tensor([[2.0, 14.0]], device="cuda:0").expand(256, 2)
    with torch.no_grad():
        dv = (agent.critic(f16, pos, goal) - agent.critic(frames, pos, goal)).abs().max().item()
    assert dv > 1e-5, dv                                                            # beyond north_star's loss tolerance
    eng.close()


def _canon_state(state):
    """Zero record fields that are meaningless in the current state (positions of unspawned patrols, wall offsets before
    the drop): the sequential and the pipelined kernel keep different leftovers there (tests/test_engine_gpu.py:_canon)."""
    from twoarmy_amd._lib import FIELDS as Fd
    ty, co, rec = state
    rec = rec.copy()
    rec[rec[:, Fd["O1_VALID"]] == 0, Fd["O1X"]:Fd["O1X"] + 6] = 0
    rec[rec[:, Fd["O2_VALID"]] == 0, Fd["O2X"]:Fd["O2X"] + 8] = 0
    rec[rec[:, Fd["PONE"]] == 0, Fd["WALL_I1"]:Fd["WALL_I1"] + 2] = 0
    return ty, co, rec


def test_graph_rollout_is_a_correct_rollout():
    """VecPPOTrainer.use_graph: the whole rollout (stack gather -> actor -> HIP sampler -> engine step -> age, T times)
    replayed as ONE HIP graph.  Two runs of the actor are not bit-reproducible (MIOpen's split-K kernels), so the graph
    rollout is checked for what it must be: (a) the engine outputs are exactly what the recorded actions produce from
    the state the rollout started in (replayed through one pipelined launch of a second engine); (b) the stored
    log-probs are the actor's log-probs of the stored actions at the stored states (1e-5); (c) the actions are what
    the sampler draws from those probabilities at the stream position the device-side offset says (> 99.9 %: a
    probability that differs in its last bit can move a threshold); (d) age / sample_count bookkeeping."""
    from twoarmy_amd import ppo_ops
    from twoarmy_amd.engine import TwoarmyEngine
    from twoarmy_amd.soa.agent.PPO import PPO
    from twoarmy_amd.soa.ppo_vec import VecPPOTrainer
    N, T = 96, 24
    torch.manual_seed(21)
    eng = TwoarmyEngine(4, N, 17, seed=SEED)
    agent = PPO()
    agent.to(eng.device).use_nhwc()
    tr = VecPPOTrainer(agent, eng, rollout_steps=T, minibatch=1024)
    tr.use_graph = True
    for r in range(4):                                      # eager (warm-up), capture + replay, replay, replay
        start_state = eng.get_state()
        start_count = agent.sample_count
        tr.collect()
        torch.cuda.synchronize()
        assert (tr._graph is not None) == (r >= 1)
        assert agent.sample_count == start_count + T * N
        # (a) engine outputs
        eng2 = TwoarmyEngine(4, N, 17, seed=SEED)
        eng2.set_state(*start_state)
        out = eng2.alloc_outputs(T)
        eng2.rollout(T, out, actions=tr.action.contiguous(), autoreset=True, policy_idx=True)
        torch.cuda.synchronize()
        assert torch.equal(out["matrix"], tr.frames[4:4 + T]) and torch.equal(out["pos"], tr.pos[4:4 + T])
        assert torch.equal(out["reward"], tr.reward) and torch.equal(out["terminated"], tr.term)
        assert torch.equal(out["truncated"], tr.trunc)
        assert all(np.array_equal(u, v) for u, v in zip(_canon_state(eng.get_state()), _canon_state(eng2.get_state())))
        eng2.close()
        # (b), (c) policy side
        idx = torch.arange(T * N, device=tr.device)
        s4, p4 = tr._stacks((idx // N).int(), (idx % N).int(), after=False)
        agent.actor.eval()
        with torch.no_grad():
            probs = agent.actor_probs(s4, p4, tr.goal1.expand(T * N, 2).contiguous())
        q = probs / probs.sum(1, keepdim=True)
        lp = torch.log(q.gather(1, tr.action.view(-1, 1).long()).clamp(1.1920929e-07, 1 - 1.1920929e-07)).view(T, N)
        assert torch.allclose(lp, tr.logp, atol=1e-5)
        a_re, _ = ppo_ops.sample(probs.contiguous(), None, seed=agent.sample_seed, offset=start_count)
        assert float((a_re.view(T, N) == tr.action).float().mean()) > 0.999
        # (d)
        d = ((tr.term | tr.trunc) != 0).cpu().numpy(); age = tr.age.cpu().numpy()
        assert np.array_equal(age[1:], np.where(d, 0, age[:-1] + 1))
        tr.carry_over()
    eng.close()


def test_predicted_frames_cached_in_the_rollout_equal_recomputed_ones():
    """PPO + predictor head (configs[4]): the frames the frozen world model predicts for an acting state are kept from
    the rollout and reused by the target pass and the epochs instead of re-running encoder -> LSTM -> decoder on the
    same stacks.  Targets and update losses with the cache == without it (1e-5: only the batch composition of the
    world-model passes differs)."""
    from twoarmy_amd.engine import TwoarmyEngine
    from twoarmy_amd.soa.agent.PPO_Predictor import ppo_predictor
    from twoarmy_amd.soa.ppo_vec import VecPPOTrainer
    N, T = 32, 40
    res = []
    for cache in (True, False):
        torch.manual_seed(17)
        eng = TwoarmyEngine(4, N, 17, seed=SEED)
        agent = ppo_predictor()
        agent.K_epochs = 1
        tr = VecPPOTrainer(agent, eng, rollout_steps=T, minibatch=512, value_chunk=512)
        assert tr.cache_predictions
        tr.cache_predictions = cache
        g = torch.Generator(device="cpu").manual_seed(9)
        tr.collect(uniforms=torch.rand(T, N, generator=g).to(tr.device))
        assert tr._pred_valid == cache
        adv, target = tr.compute_targets()
        la, lv = tr.update(permutations=[torch.randperm(T * N, generator=g)])
        res.append((tr.action.clone(), adv.clone(), target.clone(), float(la), float(lv)))
        if cache:       # the cache holds what the world model predicts for the acting states' stacks
            idx = torch.arange(T * N, device=tr.device)
            s, _ = tr._stacks((idx // N).int(), (idx % N).int(), after=False)
            want = agent.pred_states(s)[0]
            assert torch.allclose(tr.pred_frames.view(T * N, 4, 289), want, atol=1e-5)
        eng.close()
    a, b = res
    assert torch.equal(a[0], b[0])
    assert torch.allclose(a[1], b[1], atol=2e-5) and torch.allclose(a[2], b[2], atol=2e-5)
    assert abs(a[3] - b[3]) < 1e-5 and abs(a[4] - b[4]) < 1e-5
